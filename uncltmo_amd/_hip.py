"""ctypes binding of libuncltmo_hip.so (the C ABI declared in include/uncltmo_hip.h).

The product path has NO CPU fallback: if the library is missing, or a kernel is asked to run on a tensor
that is not on an MI355X, the call raises.  PyTorch is used here only for device memory and streams.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNCL_HIP_LIB points measurement tools at another build of the same ABI (same-box A/B runs); the product uses the in-tree one
LIB_PATH = os.environ.get("UNCL_HIP_LIB") or os.path.join(_HERE, "libuncltmo_hip.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GELU, ACT_SIGMOID, ACT_TANH, ACT_MSIG = 0, 1, 2, 3, 4, 5, 6
SRC_PLAIN, SRC_MAXPOOL2, SRC_CONCAT_SSR, SRC_CONCAT2, SRC_IMAGE1, SRC_CONCAT_SSR_UP = 0, 1, 2, 3, 4, 5
Z_NONE, Z_GROUPS, Z_UP2X2 = 0, 1, 2
G_NUM_WEIGHTS = 26

_ERR = {-1: "UNCL_ERR_ARG (unsupported shape / dtype / mode)", -2: "UNCL_ERR_LAUNCH", -3: "UNCL_ERR_NODEVICE"}


class HipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("ksize", C.c_int), ("pad", C.c_int), ("src_mode", C.c_int),
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
        ("src0", C.c_void_p), ("src0_H", C.c_int), ("src0_W", C.c_int), ("src0_C", C.c_int),
        ("src1", C.c_void_p), ("src1_H", C.c_int), ("src1_W", C.c_int), ("src1_C", C.c_int),
        ("prev0", C.c_void_p), ("prev_ch", C.c_int),
        ("weight", C.c_void_p), ("bias", C.c_void_p), ("act", C.c_int),
        ("scale_n", C.c_void_p), ("res", C.c_void_p), ("res_batch_stride0", C.c_int),
        ("out", C.c_void_p), ("out_H", C.c_int), ("out_W", C.c_int), ("out_C", C.c_int),
        ("z_mode", C.c_int), ("groups", C.c_int),
        ("out1_w", C.c_void_p), ("out1_b", C.c_void_p), ("out1_act", C.c_int), ("out1", C.c_void_p),
        ("skip_main_store", C.c_int),
        ("pre_w", C.c_void_p), ("pre_b", C.c_void_p), ("up_w", C.c_void_p), ("up_b", C.c_void_p),
        ("tail_w", C.c_void_p), ("tail_b", C.c_void_p),
    ]


class GenWeights(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("inc0_w", C.c_void_p), ("inc0_b", C.c_void_p),
        ("w", C.c_void_p * G_NUM_WEIGHTS), ("b", C.c_void_p * G_NUM_WEIGHTS),
        ("pos_embed", C.c_void_p), ("relative_pos", C.c_void_p), ("outc_w", C.c_void_p), ("outc_b", C.c_void_p),
        ("act", C.c_int), ("last_act", C.c_int), ("norm", C.c_int),
    ]


class GenRun(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("chunk", C.c_int), ("keep_activations", C.c_int),
        ("x", C.c_void_p), ("out", C.c_void_p), ("up_x", C.c_void_p), ("knn_idx", C.c_void_p),
        ("drop_scale", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("save_preact", C.c_int), ("prev_workspace", C.c_void_p), ("clip_T", C.c_int), ("clip_t", C.c_int),
        ("x_tile_off", C.c_void_p), ("x_pitch", C.c_int), ("x_rows", C.c_int),
    ]


class GenBwd(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("x", C.c_void_p), ("x_out", C.c_void_p), ("g_out", C.c_void_p), ("up_x", C.c_void_p),
        ("g_upx", C.c_void_p), ("drop_scale", C.c_void_p), ("workspace", C.c_void_p), ("grad_workspace", C.c_void_p),
        ("grad_workspace_bytes", C.c_size_t),
        ("wd", C.c_void_p * G_NUM_WEIGHTS), ("gw", C.c_void_p * G_NUM_WEIGHTS), ("gb", C.c_void_p * G_NUM_WEIGHTS),
        ("g_inc0_w", C.c_void_p), ("g_inc0_b", C.c_void_p), ("g_outc_w", C.c_void_p), ("g_outc_b", C.c_void_p),
        ("g_pos_embed", C.c_void_p),
        ("accumulate", C.c_int), ("prev_workspace", C.c_void_p), ("carry_in", C.c_void_p), ("carry_out", C.c_void_p),
        ("ev_decoder_done", C.c_void_p), ("clip_T", C.c_int), ("clip_t", C.c_int), ("ssr_fused", C.c_int),
    ]


_lib = None
PARAM_EPOCH = [0]   # bumped by uncltmo_amd.optim.Adam after its in-place kernel update

# name -> (restype, argtypes); every symbol include/uncltmo_hip.h declares
class ColsumItem(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("blocks", C.c_int), ("C", C.c_int), ("accumulate", C.c_int),
                ("reserved", C.c_int)]


class UnpackItem(C.Structure):
    _fields_ = [("packed", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int), ("Cin", C.c_int), ("k", C.c_int),
                ("transposed", C.c_int), ("flip", C.c_int), ("accumulate", C.c_int)]


class PackItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int), ("Cin", C.c_int), ("k", C.c_int),
                ("transposed", C.c_int), ("flip", C.c_int), ("cout_order", C.c_int)]


SIGNATURES = {
    "uncl_version": (C.c_int, []),
    "uncl_device_ok": (C.c_int, []),
    "uncl_conv_igemm": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "uncl_conv3x3_pipe": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p]),
    "uncl_conv3x3_set_pc": (C.c_int, [C.c_int]),
    "uncl_conv3x3_set_flat": (C.c_int, [C.c_int]),
    "uncl_conv3x3_dgrad_ssr": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "uncl_conv3x3_flat_count": (C.c_longlong, []),
    "uncl_wgrad_set_scratch": (C.c_int, [C.c_void_p, C.c_size_t]),
    "uncl_wgrad_scratch_bytes": (C.c_size_t, []),
    "uncl_gen_set_deterministic": (C.c_int, [C.c_int]),
    "uncl_gen_set_fused_tail": (C.c_int, [C.c_int]),
    "uncl_upconv2x2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_upconv2x2_dt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_conv_wgrad": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_conv_wgrad_bias": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_unpack_conv_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p]),
    "uncl_colsum_bf16_stage": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.POINTER(ColsumItem), C.c_void_p]),
    "uncl_colsum_finish": (C.c_int, [C.POINTER(ColsumItem), C.c_int, C.c_void_p]),
    "uncl_unpack_conv_wgrads": (C.c_int, [C.POINTER(UnpackItem), C.c_int, C.c_void_p]),
    "uncl_pack_conv_weights": (C.c_int, [C.POINTER(PackItem), C.c_int, C.c_int, C.c_void_p]),
    "uncl_colsum_workspace_bytes": (C.c_size_t, [C.c_int]),
    "uncl_colsum_bf16": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_conv3x3_dgrad": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "uncl_upconv2x2_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    "uncl_upconv2x2_dgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
    "uncl_outc_backward": (C.c_int, [C.c_void_p] * 8 + [C.c_longlong, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_ssr_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_float, C.c_int, C.c_void_p]),
    "uncl_pool_backward": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p]),
    "uncl_gelu_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "uncl_gelu_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "uncl_scale_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]),
    "uncl_mask_minus": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_longlong, C.c_float, C.c_void_p]),
    "uncl_sum_samples": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p]),
    "uncl_head_handoff": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int,
                                    C.c_void_p]),
    "uncl_mix_heads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    "uncl_gen_carry_bytes": (C.c_size_t, [C.c_int]),
    "uncl_patch_d_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "uncl_patch_d_out_size": (C.c_int, [C.c_int, C.c_int]),
    "uncl_patch_d_forward": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_patch_d_train_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "uncl_patch_d_forward_train": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_patch_d_backward": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_gauss_stats_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p]),
    "uncl_gcn_maxrel_backward": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "uncl_conv_in_c1_wgrad": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]),
    "uncl_gen_backward_workspace_bytes": (C.c_size_t, [C.c_int]),
    "uncl_gen_backward_workspace_bytes_dt": (C.c_size_t, [C.c_int, C.c_int]),
    "uncl_gen_carry_bytes_dt": (C.c_size_t, [C.c_int, C.c_int]),
    "uncl_gen_backward": (C.c_int, [C.POINTER(GenWeights), C.POINTER(GenBwd), C.c_void_p]),
    "uncl_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    "uncl_conv_in_c1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p]),
    "uncl_gcn_knn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "uncl_gcn_knn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_gcn_maxrel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p]),
    "uncl_gen_layer_name": (C.c_char_p, [C.c_int]),
    "uncl_gen_set_streams": (C.c_int, [C.c_int]),
    "uncl_gen_set_fused_graph": (C.c_int, [C.c_int]),
    "uncl_gcn_set_knn_mfma": (C.c_int, [C.c_int]),
    "uncl_wgrad_set_wide": (C.c_int, [C.c_int]),
    "uncl_wgrad_set_roll": (C.c_int, [C.c_int]),
    "uncl_wgrad_set_cat": (C.c_int, [C.c_int]),
    "uncl_wgrad_set_quad": (C.c_int, [C.c_int]),
    "uncl_checked_report": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "uncl_bnorm_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "uncl_bnorm_act": (C.c_int, [C.c_void_p] * 7 + [C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "uncl_bnorm_backward": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_gen_set_bn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "uncl_gcn_tail": (C.c_int, [C.c_void_p] * 12 + [C.c_int, C.c_int, C.c_void_p]),
    "uncl_gcn_block": (C.c_int, [C.c_void_p] * 14 + [C.c_int, C.c_int, C.c_void_p]),
    "uncl_loader_resize_crop": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_loader_gray_outputs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_loader_ldr_normalize": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "uncl_percentile_lerp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_color_finish_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]),
    "uncl_to_uint8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_warp_flow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_optical_flow_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "uncl_optical_flow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uncl_rgbe_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    "uncl_rgbe_to_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_prof_enable": (C.c_int, [C.c_int, C.c_int]),
    "uncl_prof_read": (C.c_int, [C.c_void_p, C.c_int]),
    "uncl_gen_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "uncl_gen_workspace_bytes_ex": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "uncl_inorm_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "uncl_inorm_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_gen_forward": (C.c_int, [C.POINTER(GenWeights), C.POINTER(GenRun), C.c_void_p]),
    "uncl_gauss_stats_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "uncl_gauss_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p]),
    "uncl_struct_loss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "uncl_struct_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_bicubic_half": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_simple_d_workspace_bytes": (C.c_size_t, [C.c_int]),
    "uncl_simple_d_forward": (C.c_int, [C.c_void_p] * 10 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_simple_d_backward": (C.c_int, [C.c_void_p] * 16 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_cgan_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p]),
    "uncl_nce_workspace_bytes": (C.c_size_t, [C.c_int]),
    "uncl_nce_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_nce_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_nce_similarity": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_nce_similarity_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int,
                                               C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_void_p]),
    "uncl_weighted_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_weighted_sum_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "uncl_l1_to_row": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_l1_to_row_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_l1_pairs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_void_p]),
    "uncl_tmqi_naturalness": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "uncl_gauss_var_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_add_per_sample_const": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "uncl_tv_loss": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p]),
    "uncl_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "uncl_adam_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                     C.c_float, C.c_float, C.c_void_p]),
    "uncl_tmqi_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "uncl_tmqi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_tmqi_maps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "uncl_frame_workspace_bytes": (C.c_size_t, []),
    "uncl_hdr_log_gray": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "uncl_replicate_pad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
    "uncl_order_stats": (C.c_int, [C.c_void_p, C.c_longlong, C.POINTER(C.c_ulonglong), C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "uncl_color_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_float, C.c_void_p]),
    "uncl_clamp01": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "uncl_to_uint8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "uncl_tile_count": (C.c_int, [C.c_int, C.c_int]),
    "uncl_tile_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_tile_blend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "uncl_tile_offsets": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p]),
}


def lib():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError("libuncltmo_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'`; there is no CPU fallback for the hot path." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise HipError("%s failed: %s" % (what, _ERR.get(rc, rc)))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the kernels only take HBM."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("uncltmo_amd kernels need CUDA(HIP) tensors; got a %s tensor. There is no CPU path." % t.device)
    return t.data_ptr()


def dtype_code(dt):
    if dt in ("bf16", torch.bfloat16):
        return BF16
    if dt in ("fp16", "f16", "half", torch.float16):
        return F16
    if dt in ("fp32", "f32", torch.float32):
        return F32
    raise ValueError("compute dtype must be 'bf16', 'fp16' (inference) or 'fp32', got %r" % (dt,))


def torch_dtype(code):
    return torch.bfloat16 if code == BF16 else (torch.float16 if code == F16 else torch.float32)
