"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/uncltmo_hip.h declares, the
ctypes mirror covers the header, module state_dicts match the reference's key/shape contract, the product path refuses
to run without a GPU instead of falling back, and the data-parallel gradient exchange is correct under gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "uncltmo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uncl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    lib = ctypes.CDLL(ge.LIB)
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libuncltmo_hip.so does not export %s" % n
    assert lib.uncl_version() >= 1


def test_ctypes_mirror_covers_the_header():
    from uncltmo_amd import _hip
    assert sorted(_hip.SIGNATURES) == header_functions()
    _hip.lib()          # resolves every symbol with its signature


def test_struct_layouts_match_the_c_side():
    """sizeof() of the ctypes mirrors must equal the C structs (compiled here with the host compiler)."""
    from uncltmo_amd import _hip
    code = '#include <stdio.h>\n#include "uncltmo_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(uncl_conv_desc), ' \
           'sizeof(uncl_gen_weights), sizeof(uncl_gen_run), sizeof(uncl_gen_bwd), sizeof(uncl_colsum_item), ' \
           'sizeof(uncl_pack_item), sizeof(uncl_unpack_item));return 0;}\n'
    src = os.path.join(ROOT, "build", "abi_sizes.c")
    os.makedirs(os.path.dirname(src), exist_ok=True)
    open(src, "w").write(code)
    exe = src[:-2]
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
    sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_hip.ConvDesc), ctypes.sizeof(_hip.GenWeights), ctypes.sizeof(_hip.GenRun),
                     ctypes.sizeof(_hip.GenBwd), ctypes.sizeof(_hip.ColsumItem), ctypes.sizeof(_hip.PackItem),
                     ctypes.sizeof(_hip.UnpackItem)]


def test_state_dict_contract_and_no_cpu_fallback():
    from uncltmo_amd import _hip, state_spec
    from uncltmo_amd.discriminator import SimpleDiscriminator
    from uncltmo_amd.generator import UNet, UNetVideo
    args = (1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0)
    for cls in (UNet, UNetVideo):
        g = cls(*args)
        sd = g.state_dict()
        assert [(k, tuple(s)) for k, s, _ in state_spec.generator_spec()] == [(k, tuple(v.shape)) for k, v in sd.items()]
        assert sum(v.numel() for v in sd.values()) == 4941281
        assert sum(p.numel() for p in g.parameters() if p.requires_grad) == 4920545
        with pytest.raises(RuntimeError):                      # host tensor: HipError, never an eager fallback
            g(torch.zeros(1, 1, 256, 256) if cls is UNet else torch.zeros(1, 2, 1, 256, 256))
    d = SimpleDiscriminator(256, 1, 16, "none", "none", 0, 0)
    assert sum(p.numel() for p in d.parameters()) == 12373
    with pytest.raises(RuntimeError):
        d(torch.zeros(1, 1, 256, 256))
    with pytest.raises(NotImplementedError):
        UNet(1, 1, "sigmoid", 5, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0)
    # sub-set skip operators run on the four-member kernels with zero weights: with a leaky ReLU the unused sqrt(x2) member is a
    # NaN for negative skip values and NaN * 0 = NaN -- refused, never a silent NaN (the published operator accepts leakyrelu)
    for op, lf in (("original_unet", 2), ("square", 3), ("square_root", 3)):
        UNet(1, 1, "sigmoid", 4, lf, op, 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0)
        with pytest.raises(NotImplementedError, match="leakyrelu"):
            UNet(1, 1, "sigmoid", 4, lf, op, 32, 0, "unet", 0, 0, "none", "none", "leakyrelu", 1, "replicate", 2, 0)
    UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "leakyrelu", 1, "replicate", 2, 0)
    # stretch_g: the reference builds a parameter-free module it never calls (Unet_singleFrame.py:169-175): the two names it knows
    # are accepted with the state_dict of 'none', an unknown one fails like the reference's dictionary lookup
    for name in ("batchMax", "instanceMinMax"):
        gs = UNet(*(args[:12] + (name,) + args[13:]))
        assert list(gs.state_dict().keys()) == list(UNet(*args).state_dict().keys())
    with pytest.raises(KeyError):
        UNet(*(args[:12] + ("minmax",) + args[13:]))
    # a reference-format checkpoint loads with strict=True (also through a DataParallel 'module.' prefix)
    g = UNet(*args)
    from oracle.state import generator_state
    g.load_state_dict(generator_state("g0"), strict=True)


def test_round5_entry_points_refuse_bad_arguments_before_any_launch():
    """The round-5 additions to the C ABI validate their arguments on the host (no GPU here): the fused skip backward needs a
    64-channel-tile data gradient in bf16, the interleaved cout order exists for 3x3 16-bit weights only, the optical flow needs a
    workspace of its own size, and the tiling switches return the previous setting."""
    import ctypes as C
    from uncltmo_amd import _hip
    lib = _hip.lib()
    ERR_ARG = lib.uncl_conv3x3_dgrad_ssr(None, None, None, None, 0.0, 0, None)
    assert ERR_ARG != 0
    d = _hip.ConvDesc()
    d.dtype, d.ksize, d.pad, d.N, d.H, d.W, d.Cin, d.Cout = _hip.BF16, 3, 0, 1, 34, 34, 32, 96          # 96 is not 4 C with C % 16 == 0
    assert lib.uncl_conv3x3_dgrad_ssr(C.byref(d), None, None, None, 0.0, 0, None) == ERR_ARG
    d.Cout = 128                                                                                            # shape fine, tensors missing
    assert lib.uncl_conv3x3_dgrad_ssr(C.byref(d), None, None, None, 0.0, 0, None) == ERR_ARG
    it = (_hip.PackItem * 1)()
    buf = (C.c_float * 16)()
    it[0].src, it[0].dst = C.addressof(buf), C.addressof(buf)
    it[0].Cout, it[0].Cin, it[0].k, it[0].cout_order = 128, 32, 2, 1                                       # 2x2 weights have no such order
    assert lib.uncl_pack_conv_weights(it, 1, _hip.BF16, None) == ERR_ARG
    it[0].k, it[0].Cout = 3, 96                                                                            # not four members of 16 channels
    assert lib.uncl_pack_conv_weights(it, 1, _hip.BF16, None) == ERR_ARG
    it[0].Cout = 128
    assert lib.uncl_pack_conv_weights(it, 1, _hip.F32, None) == ERR_ARG                                    # fp32 packs stay tap-major
    assert lib.uncl_optical_flow_workspace_bytes(1, 100) == 0
    need = lib.uncl_optical_flow_workspace_bytes(270, 480)
    assert need >= (16 * 270 * 480 + 2 * (135 * 240 + 68 * 120 + 34 * 60 + 17 * 30)) * 4
    assert lib.uncl_optical_flow(C.addressof(buf), C.addressof(buf), 270, 480, C.addressof(buf), C.addressof(buf), need - 1, None) == ERR_ARG
    assert lib.uncl_optical_flow(None, C.addressof(buf), 270, 480, C.addressof(buf), C.addressof(buf), need, None) == ERR_ARG
    old = lib.uncl_conv3x3_set_flat(0)
    assert lib.uncl_conv3x3_set_flat(old) == 0 and lib.uncl_conv3x3_flat_count() >= 0
    # tile origins of the overlap tiler (model_save_util.py:417-441: stride 192, last tile edge-aligned), frames outermost
    T = lib.uncl_tile_count(600, 1024)
    assert T == 3 * 5
    off = (C.c_int32 * (2 * T))()
    assert lib.uncl_tile_offsets(2, 600, 1024, off) == 0
    ys, xs = [0, 192, 600 - 256], [0, 192, 384, 576, 1024 - 256]
    assert list(off) == [(f * 600 + y) * 1024 + x for f in range(2) for y in ys for x in xs]
    assert lib.uncl_tile_offsets(2, 256, 1024, off) == ERR_ARG and lib.uncl_tile_offsets(1, 600, 1024, None) == ERR_ARG
    # the similarity pair behind nce() with longer lists: every tensor, the result and the workspace are required; fp16 is refused
    pb = C.addressof(buf)
    assert lib.uncl_nce_similarity(pb, pb, None, _hip.F32, 2, 8, 4, 0, 0, 1.0, 1e-2, pb, pb, None) == ERR_ARG
    assert lib.uncl_nce_similarity(pb, pb, pb, _hip.F32, 2, 8, 4, 0, 0, 1.0, 1e-2, None, pb, None) == ERR_ARG
    assert lib.uncl_nce_similarity(pb, pb, pb, _hip.F32, 0, 8, 4, 0, 0, 1.0, 1e-2, pb, pb, None) == ERR_ARG
    assert lib.uncl_nce_similarity_backward(pb, pb, pb, _hip.F32, 2, 8, 4, 0, 0, 1.0, 1e-2, None, pb, None, None, 0, None) == ERR_ARG
    assert lib.uncl_nce_similarity_backward(pb, pb, pb, _hip.F16, 2, 8, 4, 0, 0, 1.0, 1e-2, pb, pb, None, None, 0, None) == ERR_ARG
    assert lib.uncl_nce_similarity_backward(pb, pb, pb, _hip.F32, 2, 8, 4, 0, 0, 1.0, 1e-2, pb, None, None, None, 0, None) == 0   # nothing asked for


def test_relative_pos_buffer_matches_reference_golden(golden):
    from uncltmo_amd.generator import sincos_relative_pos
    np.testing.assert_allclose(sincos_relative_pos().numpy(), golden("generator")["relative_pos"], rtol=0, atol=1e-6)


def _ddp_worker(rank, world, port, q):
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from uncltmo_amd.distributed import allreduce_gradients, shard_range
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    x = torch.randn(8, 7)
    lo, hi = shard_range(8, rank, world)
    net(x[lo:hi]).pow(2).mean().backward()
    allreduce_gradients(net.parameters(), bucket_bytes=64)      # tiny buckets: exercise the bucketing
    q.put((rank, [p.grad.clone() for p in net.parameters()]))
    td.barrier()
    td.destroy_process_group()


def test_gradient_allreduce_equals_mean_of_rank_gradients():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    x = torch.randn(8, 7)
    ref = []
    for r in range(2):
        net.zero_grad()
        net(x[4 * r:4 * r + 4]).pow(2).mean().backward()
        ref.append([p.grad.clone() for p in net.parameters()])
    mean = [(a + b) / 2 for a, b in zip(*ref)]
    for r in range(2):
        for g_, m in zip(got[r], mean):
            np.testing.assert_allclose(g_.numpy(), m.numpy(), rtol=1e-6, atol=1e-8)


# ---------------------------------------------------------------------------------------------------------------------
# GradReducer bookkeeping at world size 2 (gloo, CPU tensors): the decoder / encoder split of _GradSet.finish, one and two
# backward passes per step, a skipped step, pre-existing .grad
# ---------------------------------------------------------------------------------------------------------------------
class _FakeG(torch.nn.Module):
    """stands in for the HIP generator: parameters + the attribute DistributedOptimizer looks for"""

    def __init__(self, variant=False):
        super().__init__()
        # (variant: a configuration whose decoder parameter is a SUB-SET of the published-layout buffer the kernels write -- rows
        # [0:2] and [3:5] of the six, like a skip operator with fewer members: generator._variant_grads)
        self.dec = torch.nn.Parameter(torch.zeros(4 if variant else 6, 4))
        self.enc = torch.nn.Parameter(torch.zeros(5, 3))
        self.bias = torch.nn.Parameter(torch.zeros(7))
        self.pe = torch.nn.Parameter(torch.zeros(3, 2))
        self._packed_weights = None


class _FakePass:
    """the buffers of one backward pass, laid out like autograd._GradSet: `flat` = [encoder | decoder] weights, `small`,
    and a transposed COPY for pos_embed; launched in _GradSet.finish's order (decoder half first, from the side stream)"""

    def __init__(self, rank, seed):
        g = torch.Generator().manual_seed(1000 * seed + rank)
        self.flat = torch.randn(15 + 24, generator=g)
        self.small = torch.randn(7 + 6, generator=g)
        self.cut = 15

    def grads(self):
        return {"enc": self.flat[:15].view(5, 3), "dec": self.flat[15:].view(6, 4), "bias": self.small[:7],
                "pe": self.small[7:].view(2, 3).t().contiguous()}

    @staticmethod
    def variant_post(g):
        """published layout -> the variant's parameters (a copy: must run after the collectives)"""
        g = dict(g)
        g["dec"] = torch.cat([g["dec"][:2], g["dec"][3:5]], 0)
        return g

    def finish(self, red, post=None):
        if red.in_stream:                               # the captured-step form of _GradSet.finish: synchronous, caller's stream
            g = self.grads()
            for t in (self.flat, self.small, g["pe"]):
                red.launch_in_stream(t, self)
            red.keep(self, g, post)
            return g
        red.launch(self.flat[self.cut:], self)          # decoder half: launched while the encoder's kernels still run
        g = self.grads()                                # pos_embed's transposed copy is made before the in-place reductions
        red.launch(self.flat[:self.cut], self)
        red.launch(self.small, self)
        red.launch(g["pe"], self)
        red.keep(self, g, post)
        return g


def _reducer_worker(rank, world, port, q):
    import torch.distributed as td
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from uncltmo_amd.distributed import DistributedOptimizer
    out = {}
    # ownership: an optimizer that updates none of the module's parameters must not take module= (it would replace the
    # generator's reducer and swallow its gradients), and parameters outside the module still take the bucketed path
    other = torch.nn.Linear(3, 2)
    try:
        DistributedOptimizer(torch.optim.SGD(other.parameters(), lr=1.0), module=_FakeG())
        raise AssertionError("module= of a foreign network was accepted")
    except ValueError:
        pass
    for case in ("one_pass", "two_pass", "skipped_step", "accumulate_into_grad", "in_stream", "extra_params", "variant", "variant_in_stream"):
        variant = case.startswith("variant")
        net = _FakeG(variant)
        extra = torch.nn.Parameter(torch.zeros(4)) if case == "extra_params" else None
        opt = DistributedOptimizer(torch.optim.SGD(list(net.parameters()) + ([extra] if extra is not None else []), lr=1.0), module=net)
        red = net._grad_reducer
        assert red.active()
        red.in_stream = case in ("in_stream", "variant_in_stream")
        if extra is not None:
            extra.grad = torch.full((4,), float(rank + 1))     # mean over the two ranks: 1.5
        if case == "skipped_step":
            _FakePass(rank, 9).finish(red)
            assert red.pending() == 1
            opt.zero_grad()                             # drops the pass of the step that was never taken
            assert red.pending() == 0
        if case == "accumulate_into_grad":
            for p in net.parameters():
                p.grad = torch.ones_like(p)
        passes = [_FakePass(rank, 1)] + ([_FakePass(rank, 2)] if case in ("two_pass", "variant") else [])
        post = _FakePass.variant_post if variant else None
        for ps in passes:
            local = {k: v.clone().numpy() for k, v in (post(ps.grads()) if variant else ps.grads()).items()}
            ps.finish(red, post)
            assert all(p.grad is None for p in net.parameters()) or case == "accumulate_into_grad"
            out.setdefault(case + ".local", []).append(local)
        opt.step()
        assert red.pending() == 0
        out[case] = {k: p.detach().clone().numpy() for k, p in net.named_parameters()}     # numpy: pickled by value
        if extra is not None:
            out[case]["extra"] = extra.detach().clone().numpy()
    q.put((rank, out))
    td.barrier()
    td.destroy_process_group()


def test_grad_reducer_world2_single_pass_two_pass_and_skipped_step():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(2):
        np.testing.assert_allclose(got[r]["extra_params"]["extra"], -1.5, rtol=1e-6)
    # "variant": the reducer is handed the PUBLISHED layout and the re-layout as a function it applies after the collectives (round 6:
    # skip-operator / bilinear variants of the generator train data-parallel): the parameter is the mean of the ranks' SLICED gradients
    for case in ("one_pass", "two_pass", "skipped_step", "accumulate_into_grad", "in_stream", "extra_params", "variant", "variant_in_stream"):
        n_pass = len(got[0][case + ".local"])
        for k in ("enc", "dec", "bias", "pe"):
            mean = sum(got[r][case + ".local"][i][k] for r in range(2) for i in range(n_pass)) / 2.0
            if case == "accumulate_into_grad":
                mean = mean + 1.0
            for r in range(2):          # SGD, lr 1, from zero: parameter = -(mean gradient); identical on both ranks
                np.testing.assert_allclose(got[r][case][k], -mean, rtol=1e-6, atol=1e-7, err_msg=case + "." + k)


def _capture_mode_worker(port, q):
    import torch.distributed as td
    from uncltmo_amd.step_graph import capture_error_mode
    before = capture_error_mode()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=0, world_size=1)
    during = capture_error_mode()
    td.destroy_process_group()
    q.put((before, during, capture_error_mode()))


def test_capture_error_mode_follows_the_process_group():
    """StepGraph / TiledGraph capture in 'thread_local' error mode exactly while a process group is alive: its watchdog thread polls
    the events of earlier collectives whenever it likes, and inside a 'global'-mode capture window such a query fails -- the
    watchdog rethrows and the process dies with SIGABRT (found with rocgdb)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_capture_mode_worker, args=(29650 + os.getpid() % 200, q))
    p.start()
    got = q.get(timeout=120)
    p.join(60)
    assert got == ("global", "thread_local", "global")
