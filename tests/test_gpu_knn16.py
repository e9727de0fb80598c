"""The kNN graph of 16-bit features (DenseDilatedKnnGraph, gcn_lib/torch_edge.py:54-86,150-158) -- index work on the
default bf16 / fp16 product path (`gcn_knn_mfma_kernel`, and its inline copy in `uncl_gcn_block`).

The fp32 indices are pinned bit for bit against the reference golden (test_gpu_generator.py).  A 16-bit run selects its graph
from an MFMA Gram matrix of the RAW 16-bit rows, so here the yardstick is built from the SAME 16-bit values: the fp64 distance
matrix of those values (normalise, |a|^2 - 2ab + |b|^2, + relative_pos).  Required:
  * the nine selected distances are the nine smallest of the fp64 row, in ascending order, within 1e-6;
  * exact ties resolve to the lower node index (constant rows: every candidate ties);
  * the matrix-core kernel and the VALU kernel (uncl_gcn_set_knn_mfma(0)) pick the same indices on >= 99.9 % of the entries
    (the fraction DESIGN.md states), and wherever they differ the two candidates are within 1e-6 of each other;
  * on every row whose ten smallest fp64 distances are pairwise more than 1e-5 apart the two kernels agree on 100 % of the indices,
    and those are the fp64 top-9 in fp64 order.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import generator as OG
from uncltmo_amd import _hip, synth
from uncltmo_amd.generator import UNet

pytestmark = pytest.mark.gpu

CH, K = 256, 9
TOL = 1e-6


def run_knn(x16, rel, mfma):
    """x16 (N, n, 256) cuda 16-bit, rel (n, n) fp32 cuda or None -> int64 cpu (N, n, 9)"""
    lib = _hip.lib()
    n_s, nodes, _ = x16.shape
    idx = torch.full((n_s, nodes, K), -1, dtype=torch.int32, device="cuda")
    code = _hip.BF16 if x16.dtype == torch.bfloat16 else _hip.F16
    old = lib.uncl_gcn_set_knn_mfma(int(mfma))
    try:
        _hip.check(lib.uncl_gcn_knn(x16.data_ptr(), code, rel.data_ptr() if rel is not None else None, idx.data_ptr(), None,
                                    n_s, nodes, CH, K, None, _hip.stream_ptr()), "uncl_gcn_knn")
        torch.cuda.synchronize()
    finally:
        lib.uncl_gcn_set_knn_mfma(old)
    return idx.cpu().long()


def ref_dist64(x16, rel):
    """fp64 distance matrix of the 16-bit values themselves (torch_edge.py:9-20, :82, :156)"""
    x = x16.double().cpu()
    xn = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    sq = (xn * xn).sum(-1, keepdim=True)
    d = sq - 2.0 * xn @ xn.transpose(1, 2) + sq.transpose(1, 2)
    if rel is not None:
        d = d + rel.double().cpu()
    return d


def check_against_fp64(idx, ref, what, tol=TOL):
    n = ref.shape[-1]
    assert idx.min().item() >= 0 and idx.max().item() < n, what
    # nine distinct neighbours per row
    assert (idx.sort(-1).values.diff(dim=-1) != 0).all(), what
    got = torch.gather(ref, 2, idx)
    want = torch.topk(ref, K, dim=2, largest=False).values           # ascending
    assert (got - want).abs().max().item() < tol, (what, (got - want).abs().max().item())
    # an exact tie in the yardstick that the kernel ALSO sees as a tie must come out lower index first; in fp64 exact ties
    # between distinct candidates only arise for identical rows, which check_constant_rows covers -- here: ascending order
    assert (got.diff(dim=-1) > -tol).all(), what


def rel_pos(n):
    """a relative_pos table: the generator's own (144,144) buffer (torch_vertex.py:203-209) cut to n nodes"""
    return OG.sincos_relative_pos(256, 12).reshape(144, 144).float()[:n, :n].contiguous()


def features(kind, n_s, nodes, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "random":
        return torch.randn(n_s, nodes, CH, generator=g)
    if kind == "correlated":          # a shared component makes the distances small and close together
        return torch.randn(n_s, 1, CH, generator=g) + 0.05 * torch.randn(n_s, nodes, CH, generator=g)
    raise ValueError(kind)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nodes", [144, 100, 33])
@pytest.mark.parametrize("with_rel", [False, True])
def test_knn16_selected_distances_are_the_smallest(dt, nodes, with_rel):
    rel = rel_pos(nodes).cuda() if with_rel else None
    for kind, n_s, seed in (("random", 37, 11), ("correlated", 9, 12)):
        x16 = features(kind, n_s, nodes, seed).to(dt).cuda()
        ref = ref_dist64(x16, rel)
        # "correlated": distances of ~4e-3 come out of (1 - 2 x.y) + 1 with every term ~1, i.e. the fp32 arithmetic the
        # reference prescribes (torch_edge.py:9-20) is itself only good to a few ulps of 1 (1.2e-7 each) there: 4e-6 stated
        tol = TOL if kind == "random" else 4e-6
        for mfma in (1, 0):
            check_against_fp64(run_knn(x16, rel, mfma), ref, (kind, dt, nodes, with_rel, mfma), tol)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_knn16_mfma_equals_valu_kernel(dt):
    """index equality of the two kernels, asserted at the fraction DESIGN.md states; the rest are last-ulp swaps"""
    for nodes, with_rel in ((144, True), (144, False), (100, True)):
        rel = rel_pos(nodes).cuda() if with_rel else None
        x16 = features("random", 200, nodes, 21).to(dt).cuda()
        a, b = run_knn(x16, rel, 1), run_knn(x16, rel, 0)
        same = (a == b)
        assert same.float().mean().item() >= 0.999, (dt, nodes, with_rel, same.float().mean().item())
        ref = ref_dist64(x16, rel)
        if not same.all():
            da, db = torch.gather(ref, 2, a), torch.gather(ref, 2, b)
            assert (da - db).abs()[~same].max().item() < 2 * TOL
        # Round 6: wherever the selection is WELL SEPARATED in the fp64 yardstick -- every gap between consecutive sorted distances up
        # to the 10th neighbour exceeds 1e-5, ten times what either kernel's fp32 arithmetic can move a distance -- the two kernels must
        # pick identical indices in identical order, 100 % of such rows (bit-exact index parity with the reference itself is a
        # property of the fp32 mode; the 16-bit graph is pinned by this yardstick)
        srt = torch.sort(ref, dim=2).values[..., :K + 1]
        well = (srt.diff(dim=-1) > 1e-5).all(-1)                      # (N, nodes): the first ten distances are pairwise separated
        assert well.float().mean().item() > 0.5, "the gate below needs rows to apply to"
        assert torch.equal(a[well], b[well]), (dt, nodes, with_rel, (a[well] != b[well]).sum().item())
        assert torch.equal(a[well], torch.topk(ref, K, dim=2, largest=False).indices[well]), (dt, nodes, with_rel)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mfma", [1, 0])
def test_knn16_exact_ties_go_to_the_lower_index(dt, mfma):
    """Constant frames give position-only (identical) feature rows: every distance of a row ties exactly, so the graph is
    0..8 for every node; with relative_pos the order is relative_pos's own, and equal table entries keep index order."""
    nodes = 144
    row = torch.randn(1, 1, CH, generator=torch.Generator().manual_seed(5))
    x16 = row.expand(3, nodes, CH).contiguous().to(dt).cuda()
    idx = run_knn(x16, None, mfma)
    assert torch.equal(idx, torch.arange(K).expand(3, nodes, K)), (dt, mfma)
    rel = rel_pos(nodes)
    # quantise the table so that it has many exactly equal entries per row
    relq = (rel * 8).round() / 8
    idx = run_knn(x16, relq.cuda(), mfma)
    ref = ref_dist64(x16, relq.cuda())
    check_against_fp64(idx, ref, ("const+rel", dt, mfma))
    sel = torch.gather(relq.expand(3, nodes, nodes), 2, idx)               # table value of every selected neighbour
    # inside the selection: equal table entries -> ascending index
    eq = sel.diff(dim=-1) == 0
    assert (idx.diff(dim=-1)[eq] > 0).all(), (dt, mfma)
    # at the boundary: an unselected node with the same table value as the last selected one has a higher index
    last_v, last_i = sel[..., -1:], idx[..., -1:]
    cand = torch.arange(nodes).expand(3, nodes, nodes)
    chosen = torch.zeros(3, nodes, nodes, dtype=torch.bool).scatter_(2, idx, True)
    bad = (~chosen) & (relq.expand(3, nodes, nodes) == last_v) & (cand < last_i)
    assert not bad.any(), (dt, mfma)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_knn16_on_the_generators_own_fc1_output(dt):
    """The features the product path really feeds the kernel: fc1 of the graph block on the golden input (oracle forward up to
    the bottleneck, rounded to the compute dtype), with the generator's relative_pos; then the whole-generator graph
    (net.infer(want_knn=True): separate kernels, fused tail, and the one-launch block with its inline copy of the kernel):
    identical graphs from the three structures, and the same graph with the VALU kernel."""
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    net = UNet(1, 1, "sigmoid", 4, 4, "square_and_square_root", 32, 0, "unet", 0, 0, "none", "none", "relu", 1, "replicate", 2, 0,
               compute_dtype=dt)
    synth.fill_state_dict(net, "g0")
    net = net.cuda().eval()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    x = torch.cat([synth.hdr_frames(1, salt="gA"), synth.smooth_hdr_frames(1, salt="gB"), torch.zeros(1, 1, 256, 256),
                   torch.full((1, 1, 256, 256), 0.5)], 0)
    want = {}
    with torch.no_grad():
        OG.unet_image_forward(sd, x, want=want)
        p = "gcn.module.0."
        t = F.conv2d(want["down3"] + sd["gcn.pos_embed"], sd[p + "0.fc1.0.weight"], sd[p + "0.fc1.0.bias"])     # (4,256,12,12)
    f16 = t.reshape(4, CH, 144).transpose(1, 2).contiguous().to(tdt).cuda()
    rel = sd[p + "0.relative_pos"].reshape(144, 144).float().contiguous().cuda()
    ref = ref_dist64(f16, rel)
    a, b = run_knn(f16, rel, 1), run_knn(f16, rel, 0)
    check_against_fp64(a, ref, ("fc1", dt, "mfma"))
    check_against_fp64(b, ref, ("fc1", dt, "valu"))
    assert (a == b).float().mean().item() >= 0.999
    # whole generator: the three structures of the block (separate kernels incl. the stand-alone matrix-core kNN; fused tail;
    # the one-launch block with its INLINE copy of the kernel) share fc1's rounding points, so their graphs must be identical:
    # the inline copy is pinned through the stand-alone kernel checked above
    lib = _hip.lib()
    old = lib.uncl_gen_set_fused_graph(0)
    try:
        with torch.no_grad():
            graphs = []
            for mode in (0, 2, 1):
                lib.uncl_gen_set_fused_graph(mode)
                _, k = net.infer(x.cuda(), want_knn=True)
                graphs.append(k.cpu().long().clone())
    finally:
        lib.uncl_gen_set_fused_graph(old)
    assert torch.equal(graphs[0], graphs[1]) and torch.equal(graphs[0], graphs[2])
    # ... and with the VALU kernel in the separate-kernel structure: same graph on >= 99.9 % of the entries
    old = lib.uncl_gen_set_fused_graph(0)
    oldk = lib.uncl_gcn_set_knn_mfma(0)
    try:
        with torch.no_grad():
            _, kv = net.infer(x.cuda(), want_knn=True)
    finally:
        lib.uncl_gcn_set_knn_mfma(oldk)
        lib.uncl_gen_set_fused_graph(old)
    assert (kv.cpu().long() == graphs[0]).float().mean().item() >= 0.999


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mfma", [1, 0])
def test_knn16_non_finite_rows_still_give_indices_inside_the_graph(dt, mfma):
    """NaN / Inf features (a diverged training step) leave a row without nine comparable candidates; the consumers gather and
    scatter by these indices, so every index must still be a node of the graph -- and the finite rows keep their neighbours
    among the finite rows."""
    nodes = 144
    x = features("random", 6, nodes, 31)
    x[0, 5] = float("nan")                  # one NaN row: its distance to everyone, and everyone's distance to it, is NaN
    x[1] = float("nan")                     # a whole NaN sample
    x[2, 7, 3] = float("inf")
    x[3, :, 0] = float("nan")               # every row of the sample has one NaN channel
    x16 = x.to(dt).cuda()
    for rel in (None, rel_pos(nodes).cuda()):
        idx = run_knn(x16, rel, mfma)
        assert idx.min().item() >= 0 and idx.max().item() < nodes, (dt, mfma, idx.min().item(), idx.max().item())
        # the untouched samples are unaffected
        ref = ref_dist64(x16[4:], rel)
        check_against_fp64(idx[4:], ref, ("finite samples", dt, mfma))
