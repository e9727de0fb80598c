/* uncltmo_hip.h — C ABI of libuncltmo_hip.so, the MI355X (gfx950) kernels behind the UnCLTMO tone-mapping
 * hot path.
 *
 * The reference (cao-cong/UnCLTMO) is pure PyTorch: it has no FFI of its own, every op on the path is a stock
 * torch.nn / torch.nn.functional call.  Each entry point below therefore cites the reference *call site* it
 * replaces (file:line in the upstream tree); the Python host in uncltmo_amd/ binds them through ctypes and
 * mirrors the reference's nn.Module / trainer signatures on top.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless its name ends in _host
 *  - activations are NHWC ("channel-last"), element type selected by `dtype` (UNCL_F32 / UNCL_BF16);
 *    accumulation is always fp32; kNN distances, loss reductions and TMQI use fp32 / fp64 as stated
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), allocates nothing, keeps no
 *    global state; scratch memory is passed in by the caller (`*_workspace_bytes` tells how much)
 *  - return value: 0 on success, a negative UNCL_ERR_* code otherwise; nothing throws
 */
#ifndef UNCLTMO_HIP_H
#define UNCLTMO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNCL_OK 0
#define UNCL_ERR_ARG (-1)      /* unsupported shape / dtype / mode */
#define UNCL_ERR_LAUNCH (-2)   /* hipLaunch reported an error      */
#define UNCL_ERR_NODEVICE (-3) /* no gfx950 device visible         */

#define UNCL_F32 0
#define UNCL_BF16 1
#define UNCL_F16 2   /* IEEE half: the inference forward (generator, tiler) only; same kernels as UNCL_BF16 on the f16 MFMA */

#define UNCL_ACT_NONE 0
#define UNCL_ACT_RELU 1
#define UNCL_ACT_LRELU 2 /* slope 0.2 */
#define UNCL_ACT_GELU 3  /* exact erf */
#define UNCL_ACT_SIGMOID 4
#define UNCL_ACT_TANH 5
#define UNCL_ACT_MSIG 6  /* 1 / (1 + exp(-3 x)): the reference's "msig" last layer (models/Blocks.py:85-91)  */

#define UNCL_PACK_MAX_ITEMS 64 /* items per launch of the batched weight re-layouts (uncl_pack_conv_weights, uncl_unpack_conv_wgrads) */

/* input-side fusions of the implicit-GEMM convolution (what the loader synthesises while staging LDS) */
#define UNCL_SRC_PLAIN 0      /* x = src0                                                                  */
#define UNCL_SRC_MAXPOOL2 1   /* x = maxpool2x2(src0), floor            (unet_parts.py:212,233)           */
#define UNCL_SRC_CONCAT_SSR 2 /* x = cat[src0, up(src1), src0^2, sqrt(src0+1e-8)] (unet_parts.py:319-322)
                                 src1 is replicate-padded to src0's size   (unet_parts.py:292-298)        */
#define UNCL_SRC_CONCAT2 3    /* x = cat[src0, src1]  ("original_unet" operator, unet_parts.py:311-312)    */
#define UNCL_SRC_IMAGE1 4     /* uncl_conv3x3_pipe only: x = act(conv3x3_valid(src0; pre_w, pre_b)), the 32-channel first
                                 layer (unet_parts.py:19) recomputed from the fp32 one-channel image src0 (N, H+2, W+2)
                                 inside the loader, so inc.conv.conv's output never goes to HBM (inference).
                                 Optional: src1 = DEVICE int32[N], element offsets of the samples' (H+2) x (W+2) windows inside
                                 src0, src1_W = row pitch of src0 in pixels, src1_H = rows of src0 in all (checked builds): the
                                 samples are tiles cut out of larger frames (model_save_util.py:409-486's crops read in place) */

#define UNCL_SRC_CONCAT_SSR_UP 5 /* uncl_conv3x3_pipe only: UNCL_SRC_CONCAT_SSR with up() fused: src1 (N, H/2, W/2, 32) is the
                                 INPUT of the 2x2 stride-2 ConvTranspose2d (unet_parts.py:269,288; weights up_w, up_b), whose
                                 32-channel output is recomputed per halo tile and never goes to HBM (inference)     */

/* what blockIdx.z enumerates besides Cout tiles */
#define UNCL_Z_NONE 0
#define UNCL_Z_GROUPS 1   /* grouped 1x1 conv: z = group (gcn_lib/torch_nn.py:58, groups=4)                */
#define UNCL_Z_UP2X2 2    /* ConvTranspose2d k2 s2 as four 1x1 GEMMs scattering to (2y+dy, 2x+dx)          */

int uncl_version(void);
int uncl_device_ok(void); /* 1 if device 0 is gfx950, else 0 */

/* ------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (bf16: v_mfma_f32_32x32x16_bf16, f32: v_mfma_f32_32x32x2_f32).
 * Replaces: nn.Conv2d / nn.ConvTranspose2d (k3 s1) + bias + ReLU in unet_parts.py:19-33,98-112,149-162
 * (a stride-1 transposed 3x3 is run as a pad-2 convolution over pre-flipped weights), the 1x1 convolutions of
 * the graph block (Unet_singleFrame.py:25-31, torch_vertex.py:190-199, torch_nn.py:58), the 2x2 stride-2
 * ConvTranspose2d of `up` (unet_parts.py:269) and `outconv`+sigmoid (unet_parts.py:338-345,
 * Unet_singleFrame.py:207-209).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct uncl_conv_desc {
  int dtype;            /* element type of src/out/weights                                   */
  int ksize;            /* 3 or 1                                                            */
  int pad;              /* 0 = valid, 2 = "full" (transposed 3x3); 1x1 ignores it            */
  int src_mode;         /* UNCL_SRC_*                                                        */
  int N, H, W;          /* logical input extent seen by the conv (after pooling / concat)    */
  int Cin, Cout;        /* logical channels (per group for UNCL_Z_GROUPS)                    */
  const void* src0;     /* NHWC                                                              */
  int src0_H, src0_W, src0_C;
  const void* src1;     /* second source for the concat modes, else NULL                     */
  int src1_H, src1_W, src1_C;
  const void* prev0;    /* video recurrence: channels [0, prev_ch) of src0 are read from this tensor of the
                           previous frame instead (Unet.py:244,270), NULL if unused             */
  int prev_ch;
  const void* weight;   /* packed by uncl_pack_conv_weight: [z][tap][Cout][Cin] in `dtype` (16-bit 3x3: K-chunk-major) */
  const float* bias;    /* [Cout_total] fp32 or NULL                                         */
  int act;              /* UNCL_ACT_* applied to conv+bias                                   */
  const float* scale_n; /* optional per-sample multiplier (DropPath keep/keep_prob), [N]     */
  const void* res;      /* optional residual added after act*scale, same layout as out       */
  int res_batch_stride0;/* 1: residual is broadcast over N (pos_embed)                       */
  void* out;            /* NHWC, out_H x out_W x out_C                                       */
  int out_H, out_W, out_C;
  int z_mode;           /* UNCL_Z_*                                                          */
  int groups;           /* for UNCL_Z_GROUPS                                                 */
  /* optional fused trailing 1x1 to one channel + sigmoid (outc): needs Cout == 32             */
  const float* out1_w;  /* [Cout] fp32 or NULL                                               */
  const float* out1_b;  /* [1]                                                               */
  int out1_act;
  float* out1;          /* [N, Hout, Wout] fp32                                              */
  int skip_main_store;  /* 1: do not write `out` (inference does not need up_x)              */
  const float* pre_w;   /* UNCL_SRC_IMAGE1: first-layer weights (32,1,3,3) fp32, reference layout */
  const float* pre_b;   /* UNCL_SRC_IMAGE1: first-layer bias (32) fp32 or NULL                    */
  const void* up_w;     /* UNCL_SRC_CONCAT_SSR_UP: packed k2 s2 transposed-conv weights [4 taps][32][32] bf16 */
  const float* up_b;    /* UNCL_SRC_CONCAT_SSR_UP: its bias (32) fp32 or NULL                      */
  /* uncl_conv3x3_pipe only, with UNCL_SRC_CONCAT_SSR_UP, act RELU, out1_* set and skip_main_store = 1: the whole last decoder
   * stage as ONE launch (inference; unet_parts.py:149-162 double_conv_traspose, :338-345 outconv, Unet_singleFrame.py:207-209).
   * tail_w = the SECOND ConvTranspose2d(32, 32, 3) packed [9 taps][32][32] like `weight` (flipped: a pad-2 correlation), applied to
   * relu(this layer's output), which stays in LDS; out1 then has extent (H + 4) x (W + 4):
   *   out1 = out1_act(outc(relu(convT3x3(relu(convT3x3(cat-ssr(src0, up(src1))))))))                                       */
  const void* tail_w;
  const float* tail_b;  /* bias of the second layer (32) fp32 or NULL */
} uncl_conv_desc;

int uncl_conv_igemm(const uncl_conv_desc* d, void* stream);

/* Pipelined bf16 variant for the 3x3 layers (persistent workgroups, register prefetch, LDS-transposed stores).
 * Same descriptor (src_mode PLAIN / CONCAT_*, no z_mode, no scale_n); `pool_out`, if not NULL, also receives
 * maxpool2x2(out) as NHWC (N, Hout/2, Wout/2, Cout) — the MaxPool2d of the next encoder stage
 * (unet_parts.py:212,233) fused into the producer. */
int uncl_conv3x3_pipe(const uncl_conv_desc* d, void* pool_out, void* stream);
/* Kernel structure used by uncl_conv3x3_pipe / uncl_conv3x3_dgrad: 0 = the four-wave kernel for every layer; 1 = producer /
 * consumer workgroups (csrc/conv3x3_pc.hip: four multiplying and four or eight staging waves, one workgroup per CU, LDS
 * planes, resident weights where they fit) for the concat-source layers only; 2 (default) = for every layer they build
 * (plain sources and single-chunk layers too; measured faster on all of them); 3 = also the layers with a fused 1x1 tail
 * (measured slower there).  All give bit-identical results; the switch
 * exists for same-process A/B timing and for the tests that compare the structures.  Returns the previous setting; on = -1 only
 * queries it (the host side packs the weights of uncl_conv3x3_dgrad_ssr -- producer / consumer kernel only -- accordingly).  Round 6:
 * with 2 or 3, inference's last layer (fused 1x1 tail, main store skipped) runs on the producer / consumer kernel with the tail
 * computed from two accumulator sets (UNCL_PC_O1C=0: the four-wave form). */
int uncl_conv3x3_set_pc(int on);
/* Tiling of the 64-channel-tile 3x3 layers (forward and data gradient) whose maps fill rectangular 8 / 16 x 32-pixel tiles badly
 * -- the 24 .. 61-pixel levels of unet_parts.py:56-87, 98-112, 149-162, 311-332: 1 (default) = flat M-tiles (csrc/conv3x3_flat.hip:
 * the output pixels of a sample linearised on the pitch Wout + 2 of the padded input, 32 per M-tile, tiles of 8 / 12 / 16 M-tiles
 * across sample borders) wherever the launcher's cost figure prefers them; 0 = rectangular tiles everywhere; 2 / 3 / 4 = flat tiles
 * with that many M-tiles per multiplying wave wherever the kernel applies (tests, A/B).  Bit-identical results either way (same
 * accumulation order per output element).  Returns the previous setting; env UNCL_FLAT sets the initial one. */
int uncl_conv3x3_set_flat(int on);
/* number of launches that took the flat tiles since the library was loaded (tests; bench.py reports it) */
long long uncl_conv3x3_flat_count(void);
/* Inference, last decoder stage (up_path.3; Unet_singleFrame.py:200-209): 1 = concat + fused up-conv -> ConvT3x3 -> ConvT3x3 ->
 * outconv + last activation as ONE launch (uncl_conv_desc.tail_w), the two 32-channel maps stay in LDS (1.9 GB less HBM traffic per
 * 200 tiles); 0 (default) = two launches with the 254 x 254 x 32 map in HBM between them -- the faster launch pair, the whole
 * forward ties.  Same rounding points either way (the fused form equals uncl_conv3x3_set_pc(3) bit for bit).
 * Returns the previous setting; env UNCL_FUSE_TAIL sets the initial one. */
int uncl_gen_set_fused_tail(int on);

/* ConvTranspose2d(k2, s2) + bias, bf16, HBM-bound layout (whole output-row runs per store).
 * Replaces `self.up(x1)` in up.forward (unet_parts.py:269,288).  x: NHWC (N,H,W,C); w: packed [4][Cout][C]
 * (uncl_pack_conv_weight with transposed=1, flip=0); out: NHWC (N,2H,2W,Cout).  `prev`/`prev_ch`: the video
 * generator's channel hand-off (Unet.py:270), NULL/0 otherwise. */
int uncl_upconv2x2(const void* x, const void* prev, int prev_ch, const void* w, const float* bias, void* out, int N,
                   int H, int W, int C, int Cout, void* stream);
/* the same with the element type stated (UNCL_BF16 or, inference, UNCL_F16) */
int uncl_upconv2x2_dt(const void* x, const void* prev, int prev_ch, const void* w, const float* bias, void* out, int dtype,
                      int N, int H, int W, int C, int Cout, void* stream);

/* Weight gradient of a 3x3 / 1x1 convolution (bf16 operands, fp32 accumulation, transposing LDS reads):
 * dw_packed[tap][Cout][Cin] += sum_p gy[p][co] * X[p + tap][ci].  Descriptor: ksize, pad, src_mode (PLAIN / CONCAT_SSR),
 * N, H, W, Cin, Cout, src0/src1 (+dims); gy: (N,Hout,Wout,Cout) bf16.  dw_packed is accumulated with float atomics:
 * zero it first.  Backward of the layers of uncl_conv3x3_pipe / the graph block's 1x1 convs. */
int uncl_conv_wgrad(const uncl_conv_desc* d, const void* gy, float* dw_packed, void* stream);
/* ... with the bias gradient of a 3x3 layer from the same pass over gy: gb[co] += sum over pixels of gy[..][co] (float atomics:
 * zero it first; NULL = weights only).  Autograd of nn.Conv2d / nn.ConvTranspose2d bias (unet_parts.py:56-64). */
int uncl_conv_wgrad_bias(const uncl_conv_desc* d, const void* gy, float* dw_packed, float* gb, void* stream);
/* 3x3 layers whose Cin and Cout are multiples of 64 can run a kernel that owns 64 x 64 channel blocks (whole 128-byte lines per
 * pixel, half the bytes per MFMA, but four times the atomic traffic per pixel-range group): mode 1 (default) = the skip-concat
 * layers, where it measured faster; 2 = every eligible layer; 0 = none (A/B runs and the tests that compare the two kernels).
 * Returns the previous setting; env UNCL_WG_WIDE sets the initial one. */
int uncl_wgrad_set_wide(int on);
/* The 32 x 32 channel-pair path of the same gradients (autograd of nn.Conv2d / nn.ConvTranspose2d weights, unet_parts.py:19-33,
 * 149-162; GanTrainerImg.py:338,460): 1 (default) = the split-role kernel (four multiplying waves that hold all nine taps and
 * walk each halo row once, four staging waves, two LDS stages, one workgroup per CU) wherever a workgroup gets at least
 * UNCL_WG_ROLL_MIN (4) tiles; 2 = always; 0 = the six-wave kernel (A/B runs, parity tests between the two).  Returns the previous
 * setting; env UNCL_WG_ROLL sets the initial one. */
int uncl_wgrad_set_roll(int on);
/* Skip-concat 3x3 layers (unet_parts.py:149-162, 319-322: the weight gradient of the conv behind torch.cat([x2, x1, x2^2, sqrt])):
 * 1 (default) = one workgroup per (32-channel skip slice, 32-channel gY chunk) covers all four members -- x1, x2 and gY are read
 * once per tile, the square and the root derived in registers; 0 = the per-pair kernels.  Returns the previous setting; env
 * UNCL_WG_CAT sets the initial one. */
int uncl_wgrad_set_cat(int on);
/* Plain 3x3 layers whose Cin and Cout are multiples of 64 (same autograd as above): 1 (default) = 64 x 64 channel blocks in the
 * split-role structure (whole 128-byte pixels staged once for the four 32 x 32 quadrants of a block) where both sides have >= 128
 * channels and a workgroup gets at least UNCL_WG_QUAD_MIN (6) 8-row tiles -- the layers it measured faster on; 2 = always; 0 = the
 * per-pair kernels.  Returns the previous setting; env UNCL_WG_QUAD. */
int uncl_wgrad_set_quad(int on);
/* Checked build only (hipcc -DUNCL_CHECKED, __graft_entry__.build_checked() -> libuncltmo_hip_checked.so): the 3x3 convolution,
 * weight-gradient and 2x2 up-conv kernels (the kernels behind nn.Conv2d / nn.ConvTranspose2d of unet_parts.py:19-33, 98-112,
 * 149-162, 269 and their autograd) validate every global access against the tensors their launch was given.  out4 = {violations,
 * first offending address, source line, bytes}; reset != 0 clears the record.  The product library returns UNCL_ERR_ARG. */
int uncl_checked_report(unsigned long long* out4, int reset);
/* unet_norm='batch_norm' in TRAINING mode (unet_parts.py:20-21, 34-35, 72-73, 85-86: nn.BatchNorm2d between every 3x3 convolution and
 * its activation; weights.norm = 2): the eighteen layers' weight / bias / running_mean / running_var and the gradient slots of
 * weight / bias, fp32 device pointers in the order inc.conv.norm, inc.conv.norm1, down_path.0..3.mpconv.1.(norm, norm1),
 * up_path.0..3.conv.(norm, norm1).  uncl_gen_forward (keep_activations, one chunk) normalises with the batch statistics and updates
 * the running ones (momentum, unbiased variance); uncl_gen_backward writes the two gradients.  Thread-local; NULL gamma clears it.
 * Eval mode needs none of this: the host folds the running statistics into the convolutions. */
/* nn.BatchNorm2d in training mode + activation on one NHWC tensor, stand-alone (the same kernels uncl_gen_forward / _backward use for
 * unet_norm='batch_norm'): x (N, HW, C) in place; zhat / rstd ([N][C]) kept for the backward; running statistics updated in place
 * (momentum, unbiased variance); scratch = uncl_bnorm_scratch_bytes(N, C).  Backward: g = dL/dy times the activation derivative, in
 * place -> dL/dz; g_gamma / g_beta [C] written or accumulated. */
size_t uncl_bnorm_scratch_bytes(int N, int C);
int uncl_bnorm_act(void* x, void* zhat, float* rstd, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   float momentum, int dtype, int N, int HW, int C, float slope, void* scratch, void* stream);
int uncl_bnorm_backward(void* g, const void* zhat, const float* rstd, const float* gamma, float* g_gamma, float* g_beta, int accumulate,
                        int dtype, int N, int HW, int C, void* scratch, void* stream);
int uncl_gen_set_bn(const float* const* gamma, const float* const* beta, float* const* running_mean, float* const* running_var,
                    float momentum, float* const* g_gamma, float* const* g_beta);
/* Deterministic weight / bias gradients (bf16 pass; autograd of nn.Conv2d / nn.ConvTranspose2d parameters, GanTrainerImg.py:338,460):
 * with a scratch buffer set, uncl_conv_wgrad / uncl_conv_wgrad_bias / uncl_upconv2x2_wgrad called from THIS thread write the
 * partial sums of their pixel-range groups there and add them up in a fixed order (one extra small launch) instead of using
 * float atomics -- the same inputs give the same bits.  uncl_gen_backward sets a slice of its own gradient workspace for the
 * duration of a bf16 pass (env UNCL_BWD_DET=0: atomics).  `scratch`: device memory the caller owns until the launches have run;
 * uncl_wgrad_scratch_bytes() is enough for every generator layer at full speed, less only lowers the groups per launch;
 * NULL restores the atomics. */
int uncl_wgrad_set_scratch(void* scratch, size_t bytes);
size_t uncl_wgrad_scratch_bytes(void);
/* uncl_gen_backward in bf16: 1 = deterministic pass (the scratch above is a slice of its gradient workspace; the max-relative
 * scatter takes its gather form), 0 (default) = float atomics -- faster (N = 32 image step 7.87 vs 8.23 ms), run-to-run different
 * in the last bits.  fp32 passes are deterministic either way.  Returns the previous setting; env UNCL_BWD_DET sets the first. */
int uncl_gen_set_deterministic(int on);
/* packed fp32 gradient -> reference layout (inverse of uncl_pack_conv_weight), written or accumulated */
int uncl_unpack_conv_wgrad(const float* packed, float* dst, int Cout, int Cin, int k, int transposed, int flip,
                           int accumulate, void* stream);
/* the same for many tensors in one launch per UNCL_PACK_MAX_ITEMS items (end of a generator backward pass) */
typedef struct uncl_unpack_item {
  const float* packed;
  float* dst;
  int Cout, Cin, k, transposed, flip, accumulate;
} uncl_unpack_item;
int uncl_unpack_conv_wgrads(const uncl_unpack_item* items, int n_items, void* stream);
/* bias gradient: out[c] (+)= sum_rows x[row][c], x bf16 [rows][ld]; workspace uncl_colsum_workspace_bytes(C) */
size_t uncl_colsum_workspace_bytes(int C);
int uncl_colsum_bf16(const void* x, long long rows, int C, int ld, float* out, int accumulate, void* workspace, void* stream);
/* The same sum in two calls, for a backward pass with many bias gradients: uncl_colsum_bf16_stage writes the per-workgroup
 * partial sums of one matrix (C <= 256) into its own `partial` buffer (512*C floats) and fills `item`; one
 * uncl_colsum_finish reduces up to UNCL_COLSUM_MAX_ITEMS staged items in a single launch (same summation order). */
#define UNCL_COLSUM_MAX_ITEMS 48
typedef struct uncl_colsum_item {
  const float* partial;
  float* out;
  int blocks, C, accumulate, reserved;
} uncl_colsum_item;
int uncl_colsum_bf16_stage(const void* x, long long rows, int C, int ld, float* partial, float* out, int accumulate,
                           uncl_colsum_item* item, void* stream);
int uncl_colsum_finish(const uncl_colsum_item* items, int n_items, void* stream);

/* Data gradient of a 3x3 layer: uncl_conv3x3_pipe over re-packed weights with an identity activation; the stored
 * gradient is multiplied by the activation derivative of the layer that produced the tensor it flows into
 * (mask > 0 ? 1 : mask_slope; mask may be NULL) and optionally added to the gradient already in `out`. */
int uncl_conv3x3_dgrad(const uncl_conv_desc* d, const void* mask, float mask_slope, int accumulate, void* stream);
/* Data gradient of a skip-concat layer (unet_parts.py:149-162 on the concatenation of :319-322) with the backward of the skip
 * operator as its epilogue: `d` describes the data-gradient convolution (Cout = 4 C, weights packed with cout_order = 1, d->out
 * unused), x2 the skip tensor (N, H, W, C) of the forward.  Writes (accumulate_x2: adds to) g_x2 = (g0 + 2 x2 g2 + g3 / (2 sqrt(x2 +
 * 1e-8))) relu'(x2) and writes g_x1 = g1, both (N, H, W, C); the 4 C-channel gradient of the concatenation never reaches memory
 * (round 5: 1.8 GB per image step).  bf16, the up-sampled map must have the skip's extent (no replicate pad).  UNCL_ERR_ARG where
 * the fused form does not apply: the caller then runs uncl_conv3x3_dgrad + uncl_ssr_backward over weights in the plain order. */
int uncl_conv3x3_dgrad_ssr(const uncl_conv_desc* d, const void* x2, void* g_x2, void* g_x1, float slope, int accumulate_x2,
                           void* stream);
/* backward of uncl_upconv2x2: weight gradient (packed [4][Cout][C], zeroed by the caller) and data gradient
 * (wt = the (Cin,Cout,2,2) weight packed as a Conv2d weight: [4][Cin][Cout]) */
int uncl_upconv2x2_wgrad(const void* x, const void* gy, float* dw_packed, int N, int H, int W, int C, int Cout, void* stream);
int uncl_upconv2x2_dgrad(const void* gy, const void* wt, const void* mask, float slope, void* gx, int N, int H, int W, int Cin,
                         int Cout, void* stream);
/* element-wise pieces of the generator backward (bf16 tensors, fp32 math); see csrc/backward_kernels.hip */
int uncl_outc_backward(const float* g_out, const float* x_out, const void* g_upx, const void* up_x, const float* w, void* G_up,
                       float* gw, float* gb, long long P, int last_act, float slope, int accumulate, void* workspace,
                       void* stream);
int uncl_ssr_backward(const void* g_cat, const void* x2, void* G_x2, void* G_x1, int N, int H, int W, int C, int H1, int W1,
                      float slope, int accumulate_x2, void* stream);
int uncl_pool_backward(const void* g_pool, const void* x, void* G_x, int N, int H, int W, int C, float slope, int accumulate,
                       void* stream);
int uncl_gelu_forward(const void* z, void* h, long long n, void* stream);
int uncl_gelu_backward(const void* g_h, const void* z, void* g_z, long long n, void* stream);
int uncl_scale_rows(const void* x, const float* scale, void* y, int N, long long per, void* stream);
int uncl_mask_minus(const void* g, const void* x, const void* pe, void* out, int N, long long per, float slope, void* stream);
int uncl_sum_samples(const void* g, float* out, int N, long long per, int accumulate, void* stream);
/* Recurrent hand-off of the video generator (Unet.py:244,270), backward side.  g: bf16 (npix, C) gradient of the MIXED input
 * of a down / up stage.  For c < prev_ch (= C/32 <= 8): carry_out[pix][c] = g[pix][c], g[pix][c] = 0 (the head gradient
 * belongs to the previous frame; carry_out NULL on the first frame) and then g[pix][c] += carry_in[pix][c] (what the next
 * frame's consumer sent back; NULL on the last frame).  If mask != NULL every channel is then multiplied by the activation
 * derivative of the layer that produced this frame's tensor (mask > 0 ? 1 : slope).  Carries are bf16 (npix, prev_ch). */
int uncl_head_handoff(void* g, const void* mask, float slope, const void* carry_in, void* carry_out, long long npix, int C,
                      int prev_ch, void* stream);
/* out = x with the first prev_ch channels of every pixel taken from prev: the mixed tensor a stage of frame k > 0 read */
int uncl_mix_heads(const void* x, const void* prev, void* out, long long npix, int C, int prev_ch, void* stream);
/* Backward of the max-relative graph convolution (gcn_lib/torch_vertex.py:22-29; out[2c] = x_c, out[2c+1] = max_k (x_c[nbr_k] - x_c)).
 * g_out: bf16 (N, n, 2C); x: bf16 (N, n, C); idx: (N, n, k) neighbour indices of the forward pass; g_x_bf16: bf16 (N, n, C) result.
 * g_x_f32: N n C floats of scratch, used (and zeroed) only by the global-atomic form, i.e. when C % 32 != 0 or a sample's slice does
 * not fit 64 KB of LDS. */
int uncl_gcn_maxrel_backward(const void* g_out, const void* x, const int32_t* idx, float* g_x_f32, void* g_x_bf16, int N, int n,
                             int C, int k, void* stream);
int uncl_conv_in_c1_wgrad(const void* G, const float* x, float* gw, float* gb, int N, int H, int W, int accumulate,
                          void* workspace, void* stream);

/* Re-layout one reference-format weight for uncl_conv_igemm.
 * src: fp32, Conv2d layout (Cout, Cin, k, k) or, if transposed != 0, ConvTranspose2d layout (Cin, Cout, k, k).
 * dst: [tap][Cout][Cin] in dtype; for a transposed stride-1 3x3 the taps are flipped (tap' = 8 - tap) so that
 * the kernel runs it as a pad-2 correlation; for the stride-2 2x2 the four taps index (dy, dx) unflipped.
 * 3x3 weights in a 16-bit dtype with Cin a multiple of 32 are stored K-CHUNK-MAJOR instead, [Cin / 32][9][Cout][32]: the MFMA
 * kernels stage one 32-channel K-chunk of all nine taps at a time, and this makes that chunk whole 128-byte lines (round 5:
 * half the L2 -> L1 requests of every layer that streams its weights).  Only this function writes the layout and only the 3x3
 * kernels read it; a caller never indexes a packed weight. */
int uncl_pack_conv_weight(const float* src, void* dst, int dtype, int Cout, int Cin, int k, int transposed,
                          int flip, void* stream);
/* The same re-layout for many weights in one launch per UNCL_PACK_MAX_ITEMS items (a training step re-packs every weight
 * of the generator, forward and data-gradient forms, after each optimiser step). */
typedef struct uncl_pack_item {
  const float* src;
  void* dst;
  int Cout, Cin, k, transposed, flip;
  int cout_order;   /* 0: as they come; 1 (3x3, 16-bit, Cout = 4 C, C a multiple of 16): the data-gradient weight of a skip-concat
                       layer for uncl_conv3x3_dgrad_ssr -- output channel g C + c (member g of [x2 | x1 | x2^2 | sqrt], channel c) is
                       stored at position 64 (c / 16) + 16 g + c % 16, so that a 64-channel tile holds the four members of 16 channels */
} uncl_pack_item;
int uncl_pack_conv_weights(const uncl_pack_item* items, int n_items, int dtype, void* stream);

/* First generator layer: Conv2d(1 -> Cout, 3x3, valid) + bias + act, input fp32 (N,H,W), output NHWC.
 * Replaces inc.conv.conv (unet_parts.py:19,26).  weight: fp32 (Cout,1,3,3). */
int uncl_conv_in_c1(const float* x, const float* w, const float* b, void* out, int dtype, int N, int H, int W,
                    int Cout, int act, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Graph block (ViG max-relative conv on the 12x12 bottleneck).
 * ---------------------------------------------------------------------------------------------------- */
/* kNN graph: x (N, n, C) NHWC features; L2-normalise over C in fp32, d_ij = |xi|^2 - 2 xi.xj + |xj|^2 +
 * relative_pos[i][j]; idx = the k smallest d per row, ascending (ties: lower j first).
 * Replaces DenseDilatedKnnGraph.forward / dense_knn_matrix (torch_edge.py:150-158, :54-86).
 * workspace: uncl_gcn_knn_workspace_bytes(N, n, C). */
size_t uncl_gcn_knn_workspace_bytes(int N, int n, int C);
int uncl_gcn_knn(const void* x, int dtype, const float* relative_pos, int32_t* idx, float* dist_out /*nullable*/,
                 int N, int n, int C, int k, void* workspace, void* stream);
/* 16-bit features without dist_out (k = 9, n > 32) take the matrix-core kernel (Gram matrix of the raw rows by MFMA, scaled by
 * 1/|x_i| 1/|x_j|, top-9 per lane); uncl_gcn_set_knn_mfma(0) selects the VALU kernel for them as well (A/B runs and the parity
 * tests that compare the two; env UNCL_KNN_MFMA sets the initial value).  Returns the previous setting. */
int uncl_gcn_set_knn_mfma(int on);
/* out (N, n, 2C): out[.., 2c] = x_c, out[.., 2c+1] = max_k (x_c[idx] - x_c).  Replaces MRConv2d.forward
 * up to its 1x1 conv (torch_vertex.py:22-29). */
int uncl_gcn_maxrel(const void* x, const int32_t* idx, void* out, int dtype, int N, int n, int C, int k,
                    void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Whole-generator forward (image UNet, published topology), one call enqueues every kernel.
 * Replaces Unet_singleFrame.UNet.forward (Unet_singleFrame.py:177-213).
 * ---------------------------------------------------------------------------------------------------- */
#define UNCL_G_NUM_WEIGHTS 26
typedef struct uncl_gen_weights {
  int dtype;
  const float* inc0_w;            /* fp32 (32,1,3,3), consumed directly by uncl_conv_in_c1  */
  const float* inc0_b;            /* fp32 (32)                                              */
  const void* w[UNCL_G_NUM_WEIGHTS];     /* packed weights, order = uncl_gen_layer_name(i)  */
  const float* b[UNCL_G_NUM_WEIGHTS];    /* fp32 biases                                     */
  const void* pos_embed;          /* (144, 256) NHWC in dtype                               */
  const float* relative_pos;      /* (144,144) fp32                                         */
  const float* outc_w;            /* (32) fp32                                              */
  const float* outc_b;            /* (1) fp32                                               */
  int act;                        /* UNCL_ACT_RELU or UNCL_ACT_LRELU (generator activation) */
  int last_act;                   /* UNCL_ACT_SIGMOID / TANH / MSIG / NONE (Unet_singleFrame.py:207-212) */
  int norm;                       /* 0: none (published model); 1: nn.InstanceNorm2d between every 3x3 conv and its activation
                                     (unet_norm = 'instance_norm', unet_parts.py:20-29).  Workspaces of such a model are sized
                                     with uncl_gen_workspace_bytes_ex(..., norm = 1).                                          */
} uncl_gen_weights;

typedef struct uncl_gen_run {
  int N;                      /* number of 256x256 tiles / frames                                          */
  int chunk;                  /* tiles per pass through the layers (0 = all); sized for the Infinity Cache */
  int keep_activations;       /* 1: every tile keeps its activations in the workspace (training / video)   */
  const float* x;             /* fp32 (N,256,256)                                                          */
  float* out;                 /* fp32 (N,256,256)                                                          */
  void* up_x;                 /* NHWC (N,256,256,32) in dtype, or NULL (inference never reads it)          */
  int32_t* knn_idx;           /* optional int32 (N,144,9)                                                  */
  const float* drop_scale;    /* optional fp32 (2,N): DropPath keep/keep_prob for the two residual sites   */
  void* workspace;
  size_t workspace_bytes;
  int save_preact;            /* 1: also keep the pre-GELU values of the graph block (needed by uncl_gen_backward)    */
  const void* prev_workspace; /* video: workspace of the previous frame (same N, keep_activations=1); the
                                 first C/32 channels entering every down/up stage come from it (Unet.py:244,270) */
  /* CLIP layout (training, Unet.py:213-289 frame loop): clip_T > 0 says `workspace` is ONE workspace laid out for clip_T * N
   * samples (uncl_gen_workspace_bytes(clip_T * N, 0, dtype, 1)), every buffer (clip_T * N, ...) with the frames one behind the
   * other, and this call is frame clip_t: it owns samples [clip_t * N, (clip_t + 1) * N) of every buffer and takes its
   * hand-off channels from frame clip_t - 1's slice (prev_workspace must be NULL: it is implied).  Needs keep_activations,
   * no chunking, norm = 0.  The backward pass can then take a clip's weight gradients in ONE launch per layer over all
   * clip_T * N samples (uncl_gen_bwd.clip_T). */
  int clip_T;
  int clip_t;
  /* Tiles read in place (inference, 16-bit dtypes, norm 0, no kept activations: the configurations whose first layer is rebuilt in
   * the second layer's loader): x_tile_off != NULL says x is a stack of x_rows rows of x_pitch fp32 pixels (whole frames one
   * behind the other) and tile i's 256 x 256 window starts x_tile_off[i] pixels into it (DEVICE int32[N]; uncl_tile_offsets fills
   * it for the reference's overlap tiling) -- the tiler's gather pass and its N x 256 KB copy disappear.  UNCL_ERR_ARG where the
   * first layer is not fused (the caller gathers then). */
  const int32_t* x_tile_off;
  int x_pitch;
  int x_rows;
} uncl_gen_run;

/* Backward of uncl_gen_forward (bf16, keep_activations = 1, save_preact = 1): gradients of every generator parameter
 * from dL/dx_out (fp32) and optionally dL/dup_x (bf16 NHWC).  Weight gradients come out in the PACKED layout of the
 * forward weights ([tap][Cout][Cin] fp32; uncl_unpack_conv_wgrad converts); `wd` are the weights re-packed for the
 * data-gradient convolutions. */
typedef struct uncl_gen_bwd {
  int N;
  const float* x;            /* generator input (N,256,256)                                */
  const float* x_out;        /* forward output (N,256,256)                                 */
  const float* g_out;        /* dL/dx_out (N,256,256)                                      */
  const void* up_x;          /* forward up_x, NHWC bf16 (N,256,256,32)                     */
  const void* g_upx;         /* dL/dup_x, NHWC bf16, or NULL                               */
  const float* drop_scale;   /* the (2,N) DropPath multipliers of the forward, or NULL     */
  void* workspace;           /* the forward's workspace                                    */
  void* grad_workspace;      /* uncl_gen_backward_workspace_bytes(N)                       */
  size_t grad_workspace_bytes;
  const void* wd[UNCL_G_NUM_WEIGHTS];   /* data-gradient weights                           */
  float* gw[UNCL_G_NUM_WEIGHTS];        /* packed fp32 weight gradients, ZEROED by caller  */
  float* gb[UNCL_G_NUM_WEIGHTS];        /* bias gradients                                  */
  float* g_inc0_w; float* g_inc0_b;     /* (32,1,3,3), (32)                                */
  float* g_outc_w; float* g_outc_b;     /* (32), (1)                                       */
  float* g_pos_embed;                   /* (144,256) NHWC fp32                             */
  /* video clips (Unet.py:213-289), backward through time, frames visited last to first:                    */
  int accumulate;            /* != 0: bias / inc / outc / pos_embed gradients are added to, not overwritten   */
  const void* prev_workspace;/* forward workspace of frame k-1 (NULL for the first frame of a clip)           */
  const void* carry_in;      /* uncl_gen_carry_bytes(N): head gradients sent back by frame k+1, or NULL        */
  void* carry_out;           /* receives the head gradients of frame k-1; required iff prev_workspace != NULL */
  void* ev_decoder_done;     /* optional hipEvent_t, recorded on the stream once the weight gradients of the decoder (packed
                                weights 14..25, outc) are complete while the graph block and the encoder are still to run: a data-
                                parallel caller starts the all-reduce of that half there (uncltmo_amd/distributed.py)            */
  /* CLIP layout with DEFERRED weight gradients (GanTrainer.py:338,460 backward over a clip): clip_T > 0 says `workspace` is the
   * clip workspace of uncl_gen_run.clip_T and `grad_workspace` one arena of uncl_gen_backward_workspace_bytes(clip_T * N); the
   * call is frame clip_t (frames still visited last to first, prev_workspace NULL = implied, carries as before).  Calls for
   * frames > 0 run the data-gradient chain only and leave their activation gradients in their slice of the arena; the call
   * for frame 0 then takes the 3x3 / 2x2 weight and bias gradients and the graph block's 1x1 ones ONCE over all clip_T * N
   * samples (a frame's weight gradients depend on nothing later in the pass) -- 26 launches per clip instead of 26 per frame.
   * outc and pos_embed stay per frame.  The first layer's weight gradient is taken by that last call as well: the `x` of frame 0
   * must be the start of ONE (clip_T * N, 256, 256) array that holds the clip's frames one behind the other (the `x` of frame t
   * is its slice t). */
  int clip_T;
  int clip_t;
  int ssr_fused;             /* bit i: decoder stage i's skip-concat data-gradient weights (wd of up_path.i.conv.conv) are packed with
                                uncl_pack_item.cout_order = 1 and the stage takes uncl_conv3x3_dgrad_ssr (bf16 only)              */
} uncl_gen_bwd;
size_t uncl_gen_backward_workspace_bytes(int N);
size_t uncl_gen_carry_bytes(int N);
/* The backward pass runs in the element type of `wts`: UNCL_BF16 (training: matrix-core kernels, fp32 atomics for the weight
 * gradients) or UNCL_F32, the PARITY mode -- activations, gradients, `wd` and `g_upx` in fp32, every weight gradient summed in a
 * fixed order by plain fp32 kernels (csrc/bwd_f32.hip; deterministic, an order of magnitude slower), so that a whole trainer step
 * can be checked against the CPU oracle at fp32 tolerances.  The two sizes above are the bf16 ones; `_dt` state the type. */
size_t uncl_gen_backward_workspace_bytes_dt(int N, int dtype);
size_t uncl_gen_carry_bytes_dt(int N, int dtype);
int uncl_gen_backward(const uncl_gen_weights* wts, const uncl_gen_bwd* b, void* stream);

const char* uncl_gen_layer_name(int i); /* state_dict prefix of packed weight i, NULL past the end */

/* Measurement aid (bench.py roofline leg): record HIP events around every launch of packed-weight layer `layer`
 * inside uncl_gen_forward, on the stream the kernels run on.  layer < 0 disables.  uncl_prof_read waits for the
 * recorded launches, writes their durations (ms) to a HOST array, resets the counter and returns how many. */
int uncl_prof_enable(int layer, int max_records);
int uncl_prof_read(float* ms_host, int max_n);
size_t uncl_gen_workspace_bytes(int N, int chunk, int dtype, int keep_activations);
/* the same for a generator with uncl_gen_weights.norm = `norm` (training keeps the normalised pre-activations and 1/std) */
size_t uncl_gen_workspace_bytes_ex(int N, int chunk, int dtype, int keep_activations, int norm);
/* InstanceNorm2d (no affine, eps 1e-5) + activation of an NHWC tensor x (N, HW, C), in place, and its backward; stand-alone
 * forms of what the generator runs per layer when norm = 1.  zhat / rstd (optional) receive the normalised pre-activation and
 * 1/std [N][C] that uncl_inorm_backward needs: g (dL/dzhat, in place) -> dL/dz. */
int uncl_inorm_act(void* x, void* zhat, float* rstd, int dtype, int N, int HW, int C, float slope, void* stream);
int uncl_inorm_backward(void* g, const void* zhat, const float* rstd, int dtype, int N, int HW, int C, void* stream);
int uncl_gen_forward(const uncl_gen_weights* wts, const uncl_gen_run* run, void* stream);
/* uncl_gen_forward runs an un-chunked inference batch of >= 64 tiles as n contiguous parts on n streams (the caller's plus
 * internal ones, forked and joined with events, so the call keeps stream semantics) up to the third decoder stage: one
 * part's launches fill the ramp-down of the other's persistent grids; the last decoder stage then runs once for the whole
 * batch on the caller's stream.  n = 1 .. 4, default 2 (200 tiles, every convolution a one-workgroup-per-CU producer /
 * consumer launch: 4.86 ms on one stream, 4.67 on two, 4.78 on four; parts are never smaller than 32 tiles); 1 = the caller's
 * stream only. */
int uncl_gen_set_streams(int n);
/* Inference (16-bit, no DropPath): uncl_gen_forward runs the tail of the graph block -- max-relative gather
 * (torch_vertex.py:22-29), grouped 1x1 conv + GELU, fc2 + residual (Grapher_noBN, :181-227), FFN fc1 + GELU, fc2 + residual
 * (Unet_singleFrame.py:20-41) -- as ONE launch with every intermediate in LDS (uncl_gcn_tail) instead of a gather kernel
 * and four 1x1 convolutions; same rounding points.  uncl_gen_set_fused_graph: 2 (default) fc1 and the kNN graph as their own
 * launches + uncl_gcn_tail; 1 the whole block incl. fc1 and the kNN in one launch (uncl_gcn_block; measured no faster);
 * 0 separate kernels (A/B, tests); returns the previous setting.
 * uncl_gcn_tail: F = fc1 output (N,144,256), idx = its kNN graph (N,144,9), X4 = the block's input; weights in the packed 1x1
 * layout of uncl_pack_conv_weight ([group][Cout][Cin]); out (N,144,256). */
int uncl_gen_set_fused_graph(int on);
int uncl_gcn_tail(const void* F, const int32_t* idx, const void* X4, const void* wg, const float* bg, const void* w2, const float* b2,
                  const void* w3, const float* b3, const void* w4, const float* b4, void* out, int dtype, int N, void* stream);
/* the whole block: w1 / b1 = fc1, relative_pos (144,144) or NULL; idx_out (N,144,9) or NULL receives the kNN graph */
int uncl_gcn_block(const void* X4, const void* w1, const float* b1, const float* relative_pos, const void* wg, const float* bg,
                   const void* w2, const float* b2, const void* w3, const float* b3, const void* w4, const float* b4, int32_t* idx_out,
                   void* out, int dtype, int N, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Window statistics.
 * ---------------------------------------------------------------------------------------------------- */
/* Per (sample, channel): mean(x) and the mean of the 11x11 sigma-1.5 Gaussian local variance G*(x^2)-(G*x)^2 over
 * the 'valid' region.  x: NHWC (N,H,W,C) in dtype, C == 1 (fp32 image) or C % 8 == 0; W <= 256.  out: fp32 (N,2,C).
 * Replaces ContrastExtracter + adaptive_avg_pool2d / mean (Unet.py:101-123,274-278; Discriminator.py:50-83,122-124;
 * GanTrainerImg.py:24-56,308-313,361-367).  Deterministic two-stage reduction. */
size_t uncl_gauss_stats_workspace_bytes(int N, int H, int C);
int uncl_gauss_stats(const void* x, int dtype, float* out, int N, int H, int W, int C, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Structural loss (models/struct_loss.py:23-104): 5x5 box-normalised windows, MSE, bicubic-halved pyramid.
 * fake, hdr: fp32 (N,H,W) device.  weights_host: HOST array of `levels` pyramid weights.  loss_out: device fp32
 * scalar = sum_l w_l * MSE_l.  grad_fake (optional): d loss / d fake, multiplied by upstream[0] if upstream != NULL.
 * ---------------------------------------------------------------------------------------------------- */
size_t uncl_struct_loss_workspace_bytes(int N, int H, int W, int levels);
int uncl_struct_loss(const float* fake, const float* hdr, const float* weights_host, int levels, float* loss_out,
                     float* grad_fake, const float* upstream, int N, int H, int W, void* workspace, void* stream);
/* F.interpolate(x, scale_factor=0.5, mode='bicubic', align_corners=False) on fp32 (N,H,W) (struct_loss.py:52-53) */
int uncl_bicubic_half(const float* in, float* out, int N, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * SimpleDiscriminator (models/Discriminator.py:87-126), fp32: forward (logit + [mean, variance] feature) and
 * backward (parameter gradients and/or input gradient).  Parameters in reference layout.
 * ---------------------------------------------------------------------------------------------------- */
size_t uncl_simple_d_workspace_bytes(int N);
int uncl_simple_d_forward(const float* x, const float* w0, const float* b0, const float* w2, const float* b2,
                          const float* w4, const float* b4, const float* wl, float* out, float* fea_final, int N,
                          void* workspace, void* gs_workspace, void* stream);
int uncl_simple_d_backward(const float* x, const float* w0, const float* w2, const float* w4, const float* wl,
                           const float* g_out, const float* g_f1, const float* g_var, float* gw0, float* gb0, float* gw2,
                           float* gb2, float* gw4, float* gb4, float* gwl, float* g_x, int accumulate, int N,
                           void* workspace, void* stream);

/* PatchGAN discriminator forward (models/Discriminator.py:129-167 NLayerDiscriminator with Blocks.Conv2dBlock,
 * models/Blocks.py:6-36, norm "instance_norm"): conv(4,2,1)+bias+LeakyReLU(0.2), (n_layers-1) x [conv(4,2,1) ->
 * InstanceNorm(eps 1e-5) -> LeakyReLU], [conv(4,1,1) -> InstanceNorm -> LeakyReLU], conv(4,1,1)+bias.  fp32.
 * x: (N,H,H) one-channel frames; w: HOST array of n_layers+2 device pointers to reference-layout weights (Cout,Cin,4,4);
 * b_first (ndf), b_last (1); out: (N,Ho,Ho) with Ho = uncl_patch_d_out_size(H, n_layers) (30 for 256 / 3 layers).
 * The reference's trainers never build it; it is kept usable as a drop-in `--d_model patchD`.
 *
 * Training form: uncl_patch_d_forward_train keeps a_0 and per normalised block {a, zhat, rstd} in `arena`
 * (uncl_patch_d_train_bytes, which also covers the two gradient buffers of the backward).  uncl_patch_d_backward takes
 * g_out (N,Ho,Ho) and OVERWRITES gw[i] (reference layout, one per convolution), gb_first (ndf), gb_last (1) and, when not
 * NULL, g_x (N,H,H).  Every sum has one owner and a fixed order: results are run-to-run identical. */
size_t uncl_patch_d_workspace_bytes(int N, int H, int ndf, int n_layers);
int uncl_patch_d_out_size(int H, int n_layers);
int uncl_patch_d_forward(const float* x, const float* const* w, const float* b_first, const float* b_last, float* out, int N, int H,
                         int ndf, int n_layers, void* workspace, void* stream);
size_t uncl_patch_d_train_bytes(int N, int H, int ndf, int n_layers);
int uncl_patch_d_forward_train(const float* x, const float* const* w, const float* b_first, const float* b_last, float* out, int N,
                               int H, int ndf, int n_layers, void* arena, void* stream);
int uncl_patch_d_backward(const float* x, const float* const* w, const float* g_out, float* const* gw, float* gb_first,
                          float* gb_last, float* g_x, int N, int H, int ndf, int n_layers, void* arena, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Loss heads.  Each returns the weighted loss (written or accumulated into a device fp32 scalar) and, where
 * pointers are given, its input gradients.
 * ---------------------------------------------------------------------------------------------------- */
/* w * (half(real,fake) + half(-fake,-real)), half = cross-entropy of [t1_i, t2_*] vs class 0 (GanTrainerImg.py:219-229) */
int uncl_cgan_loss(const float* real, const float* fake, int N, float w, float* loss, float* g_real, float* g_fake,
                   int accumulate_loss, void* stream);
/* w * mean_n CE([s(a,p), s(a,q)], 0), s(a,b) = mean_hw sum_c a b / (c + k|a-b|) (GanTrainerImg.py:410-439); pos/neg may be
 * one row shared by all samples (infoNCE2, :401-402).  E = elements per sample, hw = spatial positions.
 * accumulate_loss: bit 0 = add to *loss instead of overwriting it; bit 1 = the LMCL form (lmcl_loss, :441-450) instead of the
 * cross-entropy: w * mean_n (s(a,q) - s(a,p)), i.e. -log(exp(s_pos) / exp(s_neg)) with its one negative. */
size_t uncl_nce_workspace_bytes(int N);
int uncl_nce_loss(const void* anchor, const void* pos, const void* neg, int dtype, int N, long long E, int hw, int pos_shared,
                  int neg_shared, float k, float c, float w, float* loss, float* g_anchor, float* g_pos, float* g_neg,
                  int accumulate_loss, int accumulate_grad, void* workspace, const int* shared_rows, void* stream);
/* Gradients of uncl_nce_loss computed at backward time (autograd of GanTrainerImg.py:410-439): `workspace` is the one the
 * forward call filled, `upstream` an optional device scalar multiplied in, gradients are written in grad_dtype (= dtype, or
 * UNCL_F32).  pos_row / neg_row >= 0: the shared positive / negative is that row of `anchor` (infoNCE2, :398-402) and its
 * gradient is folded into g_anchor's row (g_pos / g_neg NULL); -1 otherwise.  E must be a multiple of 16 bytes of elements.
 * shared_rows (both calls, optional): DEVICE int[2] = {positive row, negative row} of `anchor`, e.g. the arg-max / arg-min
 * uncl_tmqi_naturalness wrote -- the selection of infoNCE2 (GanTrainerImg.py:398-402) then never visits the host. */
int uncl_nce_backward(const void* anchor, const void* pos, const void* neg, int dtype, int N, long long E, int hw, int pos_shared,
                      int neg_shared, float k, float c, const void* workspace, const float* upstream, void* g_anchor, void* g_pos,
                      void* g_neg, int grad_dtype, int pos_row, int neg_row, const int* shared_rows, void* stream);
/* The similarity on its own, for the forms of nce() the fused pair does not cover -- several positives and / or negatives per
 * anchor (GanTrainerImg.py:410-439 loops over both lists): sims[n] = {s(a_n, pos_n), s(a_n, neg_n)} (device fp32, N x 2), same
 * workspace size as uncl_nce_loss; the caller assembles its logits (InfoNCE :431-433, LMCL :441-450) from the columns.  The backward
 * takes g_sims[n] = {dL/ds_pos, dL/ds_neg} and writes (accumulate = 0) or adds fp32 gradients of the tensors whose pointer is given;
 * a row shared by all samples receives the sum. */
int uncl_nce_similarity(const void* anchor, const void* pos, const void* neg, int dtype, int N, long long E, int hw, int pos_shared,
                        int neg_shared, float k, float c, float* sims, void* workspace, void* stream);
int uncl_nce_similarity_backward(const void* anchor, const void* pos, const void* neg, int dtype, int N, long long E, int hw,
                                 int pos_shared, int neg_shared, float k, float c, const float* g_sims, float* g_anchor,
                                 float* g_pos, float* g_neg, int accumulate, void* stream);
/* err = sum_i weights[i] * terms[i][0] over device scalars, and its backward out[i] = g[0] * weights[i]: the weighting of
 * the loss terms in update_g_d_loss / train_G (GanTrainerImg.py:285-313,330-339) as one launch per direction.
 * `terms` is a HOST array of n device pointers, `weights` a host array, n <= UNCL_WSUM_MAX. */
#define UNCL_WSUM_MAX 16
int uncl_weighted_sum(const float* const* terms, const float* weights, int n, float* out, void* stream);
int uncl_weighted_sum_backward(const float* g, const float* weights, int n, float* out, void* stream);
/* w * mean_n |a_n - b_n| over strided per-sample scalars (nn.L1Loss on per-frame means, GanTrainerImg.py:308-313) */
int uncl_l1_pairs(const float* a, int a_stride, const float* b, int b_stride, int N, float w, float* loss, float* g_a,
                  float* g_b, int accumulate_loss, void* stream);
/* The two L1 terms of pseudo_label_loss (GanTrainerImg.py:360-367) in one launch.  stats: (N,2) fp32, row i = {mean, mean Gaussian
 * local variance} of patch i (uncl_gauss_stats with C = 1); row: DEVICE int, the pseudo label's patch (the arg-max
 * uncl_tmqi_naturalness wrote).  loss2[j] = mean_i |stats[i][j] - stats[row][j]|; grad (N,2) = d loss2[j] / d stats[i][j], the label
 * row receiving minus the sum of the others' signs / N (nn.L1Loss against the expanded row).  The backward scales the columns by
 * the two upstream device scalars into out (2,N): [0] per-patch gradients of the means, [1] of the local variances. */
int uncl_l1_to_row(const float* stats, int N, const int* row, float* loss2, float* grad, void* stream);
int uncl_l1_to_row_backward(const float* grad, int N, const float* g_mean_term, const float* g_var_term, float* out, void* stream);
/* TMQI statistical naturalness in fp64 (TMQI.py:210-242) of every h x w patch of fp32 frames scaled by `scale` (255);
 * best_worst (optional int32[2]): first arg-max / arg-min (GanTrainerImg.py:357-359, 398-402) */
int uncl_tmqi_naturalness(const float* x, int F, int frame_h, int frame_w, int h, int w, float scale, double* scores,
                          int32_t* best_worst, void* stream);
/* Full TMQI in fp64 (TMQI.py:107-207, `original` branch): structural fidelity over the 5-level pyramid, naturalness, and
 * Q = 0.8012 S^0.3046 + 0.1988 N^0.7088.  hdr: fp32 (H,W) luminance in any range; ldr: fp32 (H,W), multiplied by ldr_scale
 * (255 for images in [0,1]).  out: 8 doubles on the device: Q, S, N, s_local[0..4].  H, W >= 176.
 * Used by the reference's evaluators (Tester.py:339, TesterImg.py:335), not by the training step. */
size_t uncl_tmqi_workspace_bytes(int H, int W);
int uncl_tmqi(const float* hdr, const float* ldr, int H, int W, float ldr_scale, double* out, void* workspace, void* stream);
/* the same, also writing the per-level structural-fidelity maps the reference returns as `s_maps` (TMQI.py:152-157, 203-205):
 * s_maps = HOST array of five device pointers, level l receiving (H_l - 10) x (W_l - 10) doubles row-major, H_l = H >> l, W_l = W >> l;
 * NULL = uncl_tmqi */
int uncl_tmqi_maps(const float* hdr, const float* ldr, int H, int W, float ldr_scale, double* out, double* const* s_maps,
                   void* workspace, void* stream);
/* gx[n] (+)= gscale[n] * d mean(Gaussian local variance of x[n]) / dx */
int uncl_gauss_var_backward(const float* x, const float* gscale, float* gx, int N, int H, int W, int accumulate, void* stream);
/* Backward of uncl_gauss_stats for NHWC tensors (the video generator's per-frame features, Unet.py:274-278):
 * gx[n,y,x,c] (+)= g_stats[n][0][c] / (H*W) + g_stats[n][1][c] * d mean(local variance) / dx.  x, gx in dtype. */
int uncl_gauss_stats_backward(const void* x, int dtype, const float* g_stats, void* gx, int N, int H, int W, int C, int accumulate,
                              void* stream);
int uncl_add_per_sample_const(float* g, const float* scale, long long per, int N, float mul, int accumulate, void* stream);
/* w * L_TV(x) (GanTrainer.py:669-682) and its gradient; workspace: 1024 floats */
int uncl_tv_loss(const float* x, int N, int H, int W, float w, float* loss, float* gx, int accumulate_loss,
                 int accumulate_grad, void* workspace, void* stream);
/* torch.optim.Adam step over `count` tensors (HOST arrays of device pointers) */
int uncl_adam_step(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq, const int* numel,
                   int count, float lr, float beta1, float beta2, float eps, int step, void* stream);
/* the same update with the learning rate and the step count in DEVICE memory (hyper = float[2] {lr, step}, step already
 * advanced on the stream): no kernel argument changes from step to step, so a captured optimisation step (hipGraph) replays
 * with the right bias corrections.  torch.optim.Adam.step, main_train_image.py:29-32. */
int uncl_adam_step_dev(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq, const int* numel,
                       int count, const float* hyper, float beta1, float beta2, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Overlap-tile inference (256^2 tiles, stride 192, linear cross-fade).
 * Replaces test_big_size_image2 / test_big_size_image (utils/model_save_util.py:409-486, :488-565).
 * ---------------------------------------------------------------------------------------------------- */
int uncl_tile_count(int H, int W);
/* frames: fp32 (F,H,W).  tiles: fp32 (F*T, 256, 256), tile t of frame f at index f*T + t (row-major). */
int uncl_tile_gather(const float* frames, float* tiles, int F, int H, int W, void* stream);
int uncl_tile_blend(const float* tiles, float* frames, int F, int H, int W, void* stream);
/* offsets_host[f * T + t] (HOST int32[F * uncl_tile_count(H, W)], no GPU work): where tile t of frame f starts inside the stack of
 * frames, in pixels, in uncl_tile_gather's tile order -- for uncl_gen_run.x_tile_off (tiles read in place, no gather pass) */
int uncl_tile_offsets(int F, int H, int W, int32_t* offsets_host);

/* ------------------------------------------------------------------------------------------------------
 * Inference pre- / post-processing either side of the tiler, fp32 planes.
 * ---------------------------------------------------------------------------------------------------- */
/* one scratch area for the calls below */
size_t uncl_frame_workspace_bytes(void);
/* Radiance .hdr (RGBE) input (replaces imageio's FreeImage reader behind hdr_image_util.read_hdr_image,
 * utils/hdr_image_util.py:35-39).  uncl_rgbe_decode: HOST function, run-length / flat scanline decode of the bytes after
 * the resolution line into H*W*4 RGBE bytes.  uncl_rgbe_to_planes: device kernel, RGBE bytes -> (3, H/scale, W/scale) fp32
 * (value = mantissa * 2^(E-136)); scale 1, or any factor s > 1 reproducing cv2.resize(img, (W//s, H//s)) (INTER_LINEAR) of
 * load_inference2 (utils/model_save_util.py:225-226) for ANY H, W: source coordinate (d + 0.5) * (W / (W//s)) - 0.5 per axis,
 * fractional weights, clamped at the borders, horizontal pass first. */
int uncl_rgbe_decode(const uint8_t* data, size_t n, int H, int W, uint8_t* out);
int uncl_rgbe_to_planes(const uint8_t* rgbe, float* out, int H, int W, int scale, void* stream);

/* load_inference / load_inference2 arithmetic (utils/model_save_util.py:209-217): rgb (3,H,W) linear radiance ->
 * rgb_out (3,H,W; NULL to skip) = rgb - min(rgb.min(), 0) and gray_log (H,W) = log10(((Y - Y.min()) / max) * f + 1),
 * normalised by its maximum, Y = 0.299 R + 0.587 G + 0.114 B (hdr_image_util.py:68-74).  stats (device, 4 floats):
 * rgb min / max, luminance min / max. */
int uncl_hdr_log_gray(const float* rgb, int H, int W, float f_factor, float* rgb_out, float* gray_log, float* stats, void* workspace,
                      void* stream);
/* GPU-side data path of the trainers: the pixel arithmetic of npy_loader (utils/ProcessedDatasetFolderImg.py:43-206,
 * utils/ProcessedDatasetFolder.py:43-236); the random choices stay on the host.
 * uncl_loader_resize_crop: src (H, W, 3) fp32 -> the patch x patch window at (yy, xx) of cv2.resize(src, (rw, rh)) (INTER_LINEAR;
 *   rh == H, rw == W: a copy) as color (3, patch, patch) and y_plane (patch, patch; NULL to skip) = y_scale * the Y row of
 *   cv2.cvtColor(RGB2YUV), 0.299 R + 0.587 G + 0.114 B; y_scale < 0 DIVIDES by -y_scale ("bugy_max_normalization":
 *   y_scale = -255, the reference's `/ 255` at :18-19 -- a division, not a multiplication by 1/255).
 * uncl_loader_gray_outputs: gray_norm = Y / Y.max(), gray_shift = Y - Y.min() from the stats uncl_hdr_log_gray wrote (:137-144).
 * uncl_loader_ldr_normalize: "max_normalization" (mode 0) / "stretch" (mode 1) of get_ldr_im (:15-24), in place. */
int uncl_loader_resize_crop(const float* src_hwc, int H, int W, int rh, int rw, int yy, int xx, int patch, float y_scale,
                            float* color_chw, float* y_plane, void* stream);
int uncl_loader_gray_outputs(const float* color_chw, int H, int W, const float* stats, float* gray_norm, float* gray_shift,
                             void* stream);
int uncl_loader_ldr_normalize(float* x, long long n, int mode, float max_stretch, float min_stretch, void* workspace, void* stream);
/* F.pad(x, (left, W1-W-left, top, H1-H-top), mode='replicate') on `planes` planes of H x W
 * (data_loader_util.add_frame_to_im / add_frame_to_im_batch / resize_im, utils/data_loader_util.py:135-185) */
int uncl_replicate_pad(const float* x, float* y, int planes, int H, int W, int top, int left, int H1, int W1, void* stream);
/* out[r] = the ranks[r]-th smallest of the n values of x (0-based, exact; radix select, no sort).  ranks: HOST array,
 * nr <= 8.  The two neighbours of a fractional rank are what np.percentile interpolates (model_save_util.py:389-390,
 * hdr_image_util.py:93-97). */
int uncl_order_stats(const float* x, long long n, const unsigned long long* ranks, int nr, float* out, void* workspace, void* stream);
/* clamp to [lo,hi], stretch to [0,1], multiply by sqrt(rgb / (Y + 1e-8)) and cut the H x W window at (top,left) out of the
 * padded H1 x W1 planes (model_save_util.py:391-400, hdr_image_util.back_to_color_tensor :120-131).  out: (3,H,W). */
int uncl_color_finish(const float* rgb, const float* fake, float* out, int H1, int W1, int top, int left, int H, int W, float lo,
                      float hi, void* stream);
int uncl_clamp01(const float* x, float* y, long long n, void* stream);
/* (C,H,W) fp32 -> (H,W,C) uint8: clamp(x,0,1), (v - lo) / (hi - lo), clip to [0,1], truncate(v * 255)
 * (hdr_image_util.save_gray_tensor_as_numpy_stretch :237-241 with to_0_1_range_outlier :93-103) */
int uncl_to_uint8(const float* x, unsigned char* out, int C, int H, int W, float lo, float hi, void* stream);
/* Device-resident variants for a frame pipeline without host round trips: uncl_percentile_lerp finishes np.percentile's
 * linear interpolation (numpy `_lerp`, every operation rounded separately, float32 or float64 as numpy's promotion gives) from
 * the order statistics of uncl_order_stats; gamma / f64 are HOST arrays (n <= 8), pairs / out device.  uncl_color_finish_dev /
 * uncl_to_uint8_dev read [lo, hi] from device memory (to_uint8: a zero span gets hi += 1e-8, hdr_image_util.py:98-99). */
int uncl_percentile_lerp(const float* pairs, const double* gamma, const int* f64, int n, float* out, void* stream);
int uncl_color_finish_dev(const float* rgb, const float* fake, float* out, int H1, int W1, int top, int left, int H, int W,
                          const float* lohi, void* stream);
int uncl_to_uint8_dev(const float* x, unsigned char* out, int C, int H, int W, const float* lohi, void* stream);
/* warp_flow (GanTrainer.py:584-595; used by Tester.eval_on_video's warp error, Tester.py:379-389): out = cv2.remap(img, flow + pixel
 * grid, None, cv2.INTER_LINEAR).  img: (H,W,C) uint8, flow: (Hf,Wf,2) fp32 displacements (x, y) -- NOT modified (the reference adds
 * the grid into its argument in place), out: (Hf,Wf,C) uint8.  OpenCV's fixed-point bilinear remap for 8-bit images restated
 * (1/32-pixel coordinates, 15-bit weights, constant-0 border); cv2 is absent here: parity with cv2 unpinned, hand-computed vectors. */
int uncl_warp_flow(const unsigned char* img, const float* flow, unsigned char* out, int H, int W, int C, int Hf, int Wf, void* stream);
/* Dense inverse optical flow between two frames for the evaluator's warp error -- the role of `compute_flow(img_to_align,
 * img_source)` / `estimate_invflow` (GanTrainer.py:597-646; callers Tester.py:379-384, metrics/compute_wrap_error.py:105-114),
 * which the reference fills with cv2.optflow DeepFlow.  OpenCV is absent from the reference tree and this image (parity with cv2
 * unpinned); this is coarse-to-fine iterative Lucas-Kanade with 15 x 15 box windows, operation for operation oracle/flow.py, which
 * is pinned by synthetic motions with known flow.  img_to_align, img_source: (H, W) fp32 planes in [0, 255] (channel 0 of the
 * frames); flow: (H, W, 2) fp32, flow[..., 0] along x, with img_to_align(p + flow(p)) ~ img_source(p): what uncl_warp_flow takes. */
size_t uncl_optical_flow_workspace_bytes(int H, int W);
int uncl_optical_flow(const float* img_to_align, const float* img_source, int H, int W, float* flow, void* workspace,
                      size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNCLTMO_HIP_H */
