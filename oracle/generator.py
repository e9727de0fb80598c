"""ORACLE (test infrastructure, never shipped): CPU fp32 restatement of the UnCLTMO generators.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  The
product path (uncltmo_amd/) never does and fails loudly when its HIP library is missing.

Parity pin: this restatement is checked against golden vectors captured from the upstream reference
itself, run in the build container behind import shims (tests/golden/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_golden.py).  DropPath in train mode comes from an unpinned
third-party package (timm); it is pinned only through an explicitly injected keep-mask.

Functional style: the network is a pure function of (state_dict, input); nothing is an nn.Module.
Each function cites the reference lines it restates (paths relative to the upstream repo).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS_SSR = 1e-08  # utils/params.py:48


# --------------------------------------------------------------------------------------------
# fixed buffers
# --------------------------------------------------------------------------------------------
def sincos_relative_pos(embed_dim=256, grid=12):
    """-(2 * P P^T / D) for the 2-D sin/cos position table P, as float32 (1, n, n).

    gcn_lib/pos_embed.py:21-29 (relative), :38-83 (sincos table; note the meshgrid puts w first so
    the first half of the embedding encodes the column index), gcn_lib/torch_vertex.py:203-209
    (negation; the bicubic resize to (n, n) is the identity at r=1).
    """
    half = embed_dim // 2
    omega = np.arange(half // 2, dtype=np.float64) / (half / 2.0)
    omega = 1.0 / 10000 ** omega
    gy, gx = np.meshgrid(np.arange(grid, dtype=np.float32), np.arange(grid, dtype=np.float32), indexing="ij")

    def enc(pos):
        out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    table = np.concatenate([enc(gx), enc(gy)], axis=1)          # (n, D): column index first
    rel = 2.0 * table @ table.T / table.shape[1]
    rel32 = torch.from_numpy(np.float32(rel)).unsqueeze(0).unsqueeze(1)
    n = grid * grid
    rel32 = F.interpolate(rel32, size=(n, n), mode="bicubic", align_corners=False)
    return -rel32.squeeze(1)


def gauss_window(size=11, sigma=1.5):
    """Normalised 2-D Gaussian, float32 (1,1,size,size).  Unet.py:101-106, Discriminator.py:50-55."""
    ax = np.arange(-size // 2 + 1, size // 2 + 1)
    x, y = np.meshgrid(ax, ax, indexing="ij")
    g = np.exp(-((x ** 2 + y ** 2) / (2.0 * sigma ** 2)))
    return torch.from_numpy(g / g.sum()).float().unsqueeze(0).unsqueeze(0)


def local_variance(x, win=None):
    """Gaussian-weighted local variance per channel, 'valid' borders.  Unet.py:112-123."""
    if win is None:
        win = gauss_window()
    b, c, h, w = x.shape
    xr = x.reshape(b * c, 1, h, w)
    win = win.to(x.dtype)           # the fp64 evaluation used as the ground truth of the fp32 parity tests
    mu = F.conv2d(xr, win)
    var = F.conv2d(xr * xr, win) - mu.pow(2)
    return var.reshape(b, c, var.shape[2], var.shape[3])


# --------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------
def _act(x, activation):
    if activation == "relu":
        return F.relu(x)
    if activation == "leakyrelu":
        return F.leaky_relu(x, 0.2)
    raise AssertionError("Unsupported activation: {%s}" % activation)


def _norm(x, unet_norm, sd=None, q=None, training=False):
    """unet_parts.py:20-29, 34-37, 72-73, 85-86: the norm between a 3x3 convolution and its activation.  `q` = the state_dict prefix
    of the norm module ("inc.conv.norm", "...norm1").  nn.BatchNorm2d defaults: affine, eps 1e-5, momentum 0.1; in training mode
    F.batch_norm normalises with the batch statistics and updates sd's running statistics in place, like the module."""
    if unet_norm in (None, "none"):
        return x
    if unet_norm == "instance_norm":
        return F.instance_norm(x, eps=1e-5)          # nn.InstanceNorm2d default: no affine, no stats
    if unet_norm == "batch_norm":
        if training and (q + ".num_batches_tracked") in sd:
            sd[q + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"],
                            training=training, momentum=0.1, eps=1e-5)
    raise NotImplementedError("oracle covers unet_norm in {none, instance_norm, batch_norm}")


def double_conv(sd, p, x, first_transposed, second_transposed, activation="relu", unet_norm="none", training=False):
    """conv -> [norm] -> act -> conv1 -> [norm] -> act with no padding.

    unet_parts.py:56-87 (double_conv, valid 3x3), :126-141 (double_last_conv: conv then ConvT),
    :180-193 (double_conv_traspose: ConvT, ConvT).  Transposed 3x3 stride 1 grows the map by 2.
    """
    f0 = F.conv_transpose2d if first_transposed else F.conv2d
    f1 = F.conv_transpose2d if second_transposed else F.conv2d
    x = _act(_norm(f0(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"]), unet_norm, sd, p + ".norm", training), activation)
    x = _act(_norm(f1(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"]), unet_norm, sd, p + ".norm1", training), activation)
    return x


def skip_concat(x2, x1, con_operator="square_and_square_root"):
    """unet_parts.py:311-332: channel concat of the skip x2 with the upsampled x1."""
    if con_operator == "original_unet":
        return torch.cat([x2, x1], dim=1)
    if con_operator == "square":
        return torch.cat([x2, x1, torch.pow(x2, 2)], dim=1)
    if con_operator == "square_root":
        return torch.cat([x2, x1, torch.pow(x2 + EPS_SSR, 0.5)], dim=1)
    if con_operator == "square_and_square_root":
        return torch.cat([x2, x1, torch.pow(x2, 2), torch.pow(x2 + EPS_SSR, 0.5)], dim=1)
    if con_operator == "gamma":
        return torch.cat([x2, x1, torch.pow(x2 + EPS_SSR, 0.02)], dim=1)
    raise AssertionError("Unsupported con_operator request: {}".format(con_operator))


def up_block(sd, p, x1, x2, con_operator, activation="relu", unet_norm="none", training=False):
    """ConvT 2x2 stride 2, replicate-pad x1 up to x2's size (right/bottom get the odd pixel), concat,
    double transposed conv.  unet_parts.py:283-335 (pad :292-298)."""
    if p + ".up.weight" not in sd and p + ".up.1.weight" not in sd:
        # up_mode=1 (unet_parts.py:284-288): zero-insertion upsampling -- x1 lands on the even pixels, zeros in between
        w = x1.new_zeros(2, 2)
        w[0, 0] = 1
        x1 = F.conv_transpose2d(x1, w.expand(x1.size(1), 1, 2, 2), stride=2, groups=x1.size(1))
    elif p + ".up.1.weight" in sd:       # bilinear=1 (unet_parts.py:256-259): nn.Upsample(scale_factor=2) [nearest] + 1x1 convolution
        x1 = F.conv2d(F.interpolate(x1, scale_factor=2), sd[p + ".up.1.weight"], sd[p + ".up.1.bias"])
    else:
        x1 = F.conv_transpose2d(x1, sd[p + ".up.weight"], sd[p + ".up.bias"], stride=2)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    if dx or dy:
        x1 = F.pad(x1, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2), mode="replicate")
    x = skip_concat(x2, x1, con_operator)
    return double_conv(sd, p + ".conv", x, True, True, activation, unet_norm, training)


# --------------------------------------------------------------------------------------------
# graph-conv bottleneck
# --------------------------------------------------------------------------------------------
def knn_graph(x, relative_pos, k=9):
    """Dense kNN over the H*W nodes of x (B,C,H,W) -> int64 (B, n, k) neighbour indices.

    gcn_lib/torch_edge.py:150-158 (L2-normalise over C), :9-20 (|a|^2 - 2ab + |b|^2), :80-83
    (+relative_pos, top-k of the negated distance).  dilation = 1 keeps all k (torch_edge.py:131).
    """
    b, c, h, w = x.shape
    with torch.no_grad():
        xn = F.normalize(x.reshape(b, c, -1, 1), p=2.0, dim=1)
        pts = xn.transpose(2, 1).squeeze(-1)                      # (B, n, C)
        inner = -2 * torch.matmul(pts, pts.transpose(2, 1))
        sq = torch.sum(pts * pts, dim=-1, keepdim=True)
        dist = sq + inner + sq.transpose(2, 1)
        if relative_pos is not None:
            dist = dist + relative_pos
        _, idx = torch.topk(-dist, k=k)
    return idx, dist


def max_relative(x, idx):
    """x (B,C,n) , idx (B,n,k) -> (B,2C,n): channels interleave [x_c, max_k(x_c[nbr] - x_c)].

    gcn_lib/torch_vertex.py:22-29 with gcn_lib/torch_nn.py:81-102 (gather).
    """
    b, c, n = x.shape
    k = idx.shape[-1]
    nb = torch.gather(x.unsqueeze(-1).expand(b, c, n, k), 2, idx.unsqueeze(1).expand(b, c, n, k))
    rel = (nb - x.unsqueeze(-1)).max(dim=-1).values
    return torch.stack([x, rel], dim=2).reshape(b, 2 * c, n)


def gcn_block(sd, x, drop_keep=None, drop_prob=0.05, training=False, want=None):
    """`x + pos_embed` -> Grapher_noBN -> FFN.  Unet_singleFrame.py:92-99, :36-42;
    gcn_lib/torch_vertex.py:217-227, :121-130.

    drop_keep: optional (2, B) 0/1 keep flags for the two DropPath sites (train mode only); the branch
    is scaled by 1/(1-drop_prob) as timm's DropPath does.  dpr = linspace(0.05, 0.1, 1)[0] = 0.05
    (Unet_singleFrame.py:62).
    """
    p = "gcn.module.0."
    x = x + sd["gcn.pos_embed"]
    b, c, h, w = x.shape

    def drop(branch, site):
        if not training or drop_prob == 0.0:
            return branch
        assert drop_keep is not None, "train-mode oracle needs an explicit DropPath keep mask"
        m = drop_keep[site].to(branch).reshape(b, 1, 1, 1) / (1.0 - drop_prob)
        return branch * m

    # Grapher
    t = F.conv2d(x, sd[p + "0.fc1.0.weight"], sd[p + "0.fc1.0.bias"])
    idx, dist = knn_graph(t, sd[p + "0.relative_pos"], k=9)
    if want is not None:
        want["knn_idx"] = idx
        want["knn_dist"] = dist
    g = max_relative(t.reshape(b, c, h * w), idx).reshape(b, 2 * c, h * w, 1)
    g = F.gelu(F.conv2d(g, sd[p + "0.graph_conv.gconv.nn.0.weight"], sd[p + "0.graph_conv.gconv.nn.0.bias"],
                        groups=4))
    g = g.reshape(b, -1, h, w)
    g = F.conv2d(g, sd[p + "0.fc2.0.weight"], sd[p + "0.fc2.0.bias"])
    x = drop(g, 0) + x
    # FFN
    f = F.gelu(F.conv2d(x, sd[p + "1.fc1.0.weight"], sd[p + "1.fc1.0.bias"]))
    f = F.conv2d(f, sd[p + "1.fc2.0.weight"], sd[p + "1.fc2.0.bias"])
    return drop(f, 1) + x


# --------------------------------------------------------------------------------------------
# generators
# --------------------------------------------------------------------------------------------
def _last_act(x, last_layer):
    if last_layer == "sigmoid":
        return torch.sigmoid(x)
    if last_layer == "tanh":
        return torch.tanh(x)
    if last_layer == "msig":
        return 1 / (1 + torch.exp(-3 * x))        # Blocks.py:85-91 with factor 3
    return x


def crop_center(x, diffY, diffX):
    """utils/data_loader_util.py:165-172."""
    b, c, h, w = x.shape
    th, tw = h - diffY, w - diffX
    i = int(round((h - th) / 2.0))
    j = int(round((w - tw) / 2.0))
    return x[:, :, i:i + th, j:j + tw]


def unet_image_forward(sd, x, con_operator="square_and_square_root", last_layer="sigmoid", activation="relu",
                       unet_norm="none", training=False, drop_keep=None, want=None,
                       to_crop=False, apply_crop=True, diffY=0, diffX=0):
    """Image generator: (N,1,256,256) -> (x_out (N,1,256,256), up_x (N,32,256,256)).

    Unet_singleFrame.py:177-213; published topology depth=4, filters=32, doubleConvTranspose=1,
    up_mode=0, convtranspose_kernel=2 (activate_trained_model/model_weights_imageTMO/run_settings.npy).
    `want`, if a dict, receives named intermediates for layer-level parity checks.
    """
    if x.shape[-1] != 256 or x.shape[-2] != 256:
        raise ValueError("generator accepts only 256x256 inputs (12x12 pos_embed); got %s" % (tuple(x.shape),))

    def rec(name, t):
        if want is not None:
            want[name] = t
        return t

    a, n = activation, unet_norm
    feats = [rec("inc", double_conv(sd, "inc.conv", x, False, False, a, n, training))]
    for i in range(3):
        feats.append(rec("down%d" % i, double_conv(sd, "down_path.%d.mpconv.1" % i, F.max_pool2d(feats[-1], 2),
                                                   False, False, a, n, training)))
    feats.append(rec("down3", double_conv(sd, "down_path.3.mpconv.1", F.max_pool2d(feats[-1], 2),
                                          False, True, a, n, training)))
    up = rec("gcn", gcn_block(sd, feats[4], drop_keep=drop_keep, training=training, want=want))
    for i in range(4):
        up = rec("up%d" % i, up_block(sd, "up_path.%d" % i, up, feats[3 - i], con_operator, a, n, training))
    out = _last_act(F.conv2d(up, sd["outc.conv.weight"], sd["outc.conv.bias"]), last_layer)
    if apply_crop and to_crop:
        out = crop_center(out, diffY, diffX)
    return out, up


def unet_video_forward(sd, x, con_operator="square_and_square_root", last_layer="sigmoid", activation="relu",
                       unet_norm="none", training=False, drop_keep=None, ratio=1.0 / 32, want=None,
                       detach_handoff=False):
    """Video generator: (B,T,1,256,256) -> (frames (B,T,1,256,256), feats (B,T,64,1,1)).

    Unet.py:213-289.  From the second frame on, the first int(C*ratio) channels of the *input* of every
    down/up stage are replaced by the same channels of the previous frame's corresponding stage output
    (not detached: gradients flow back through time).  feats = [mean(up_x), mean(local variance of
    up_x under an 11x11 sigma-1.5 Gaussian)] per channel (Unet.py:274-278).
    drop_keep, if given, is (T, 2, B).  detach_handoff is a TEST knob (not in the reference): it cuts the gradient
    through time so that a test can show how much of a gradient travels through the hand-off.
    """
    a, n = activation, unet_norm
    outs, fts, last = [], [], None
    win = gauss_window()
    for k in range(x.shape[1]):
        xf = x[:, k]
        if xf.shape[-1] != 256 or xf.shape[-2] != 256:
            raise ValueError("generator accepts only 256x256 inputs (12x12 pos_embed)")
        cur = []

        def head(t):
            return t[:, :int(t.shape[1] * ratio)]

        def mix(t, slot):
            if k == 0:
                return t
            return torch.cat((last[slot], t[:, int(t.shape[1] * ratio):]), 1)

        nx = double_conv(sd, "inc.conv", xf, False, False, a, n, training)
        feats = [nx]
        cur.append(head(nx))
        for i in range(4):
            nx = double_conv(sd, "down_path.%d.mpconv.1" % i, F.max_pool2d(mix(nx, i), 2), False, i == 3, a, n, training)
            feats.append(nx)
            cur.append(head(nx))
        dk = None if drop_keep is None else drop_keep[k]
        up = gcn_block(sd, feats[4], drop_keep=dk, training=training,
                       want=want if (want is not None and k == 0) else None)
        cur.append(head(up))
        for i in range(4):
            up = up_block(sd, "up_path.%d" % i, mix(up, 5 + i), feats[3 - i], con_operator, a, n, training)
            cur.append(head(up))
        f1 = up.mean(dim=(2, 3), keepdim=True)
        f2 = local_variance(up, win).mean(dim=(2, 3), keepdim=True)
        fts.append(torch.cat([f1, f2], dim=1).unsqueeze(1))
        outs.append(_last_act(F.conv2d(up, sd["outc.conv.weight"], sd["outc.conv.bias"]), last_layer).unsqueeze(1))
        last = [h.detach() for h in cur] if detach_handoff else cur
    return torch.cat(outs, 1), torch.cat(fts, 1)
