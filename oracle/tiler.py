"""ORACLE (test infrastructure): overlap-tile inference with linear cross-fade.

Restates utils/model_save_util.py:409-486 (4-D image tiler) and :488-565 (5-D video tiler): 256^2 patches
with stride 192; along each axis a tile that overlaps the running result is blended in over its first
`overlap` positions with weights i/(overlap-1), and a final tile aligned to the far edge is blended over
`last_range = end_of_last_regular_tile - (L - patch)` positions with weights i/(last_range-1).  Rows of
tiles are first blended horizontally into strips, then strips are blended vertically the same way.
Quirks kept: H or W <= patch is unsupported (the reference's loop variables are undefined there).
"""
import torch


def axis_plan(L, patch=256, overlap=64):
    """[(start, blend_len)] for one axis: regular tiles (blend_len = overlap, 0 for the first) and the
    final edge-aligned tile."""
    if L <= patch:
        raise ValueError("tiler needs every spatial dim > patch (reference loop is undefined otherwise)")
    plan, idx = [], 1
    while patch * idx - overlap * (idx - 1) < L:
        plan.append(((patch - overlap) * (idx - 1), 0 if idx == 1 else overlap))
        idx += 1
    end_last = plan[-1][0] + patch
    plan.append((L - patch, end_last - (L - patch)))
    return plan


def tile_count(H, W, patch=256, overlap=64):
    return len(axis_plan(H, patch, overlap)) * len(axis_plan(W, patch, overlap))


def _fold(acc, piece, start, blend, dim, patch):
    """Blend `piece` (extent `patch` along dim) into acc at `start`."""
    sl = [slice(None)] * acc.dim()
    pl = [slice(None)] * acc.dim()
    for i in range(blend):
        sl[dim], pl[dim] = start + i, i
        acc[tuple(sl)] = acc[tuple(sl)] * (blend - 1 - i) / (blend - 1) + piece[tuple(pl)] * i / (blend - 1)
    sl[dim], pl[dim] = slice(start + blend, start + patch), slice(blend, patch)
    acc[tuple(sl)] = piece[tuple(pl)]


def tiled_forward(x, model, patch=256, overlap=64, **model_kw):
    """x: (N,1,H,W) or (B,T,1,H,W); model(patch) -> (out, _) with out shaped like the patch."""
    H, W = x.shape[-2], x.shape[-1]
    ys, xs = axis_plan(H, patch, overlap), axis_plan(W, patch, overlap)
    out = torch.zeros_like(x)
    hd, wd = x.dim() - 2, x.dim() - 1
    for (y0, yb) in ys:
        strip = torch.zeros(x.shape[:-2] + (patch, W), dtype=x.dtype)
        for (x0, xb) in xs:
            with torch.no_grad():
                o, _ = model(x[..., y0:y0 + patch, x0:x0 + patch], **model_kw)
            _fold(strip, o, x0, xb, wd, patch)
        _fold(out, strip, y0, yb, hd, patch)
    return out
