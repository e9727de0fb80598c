"""state_dict layouts of the published generator / discriminator.

Key names and shapes are the checkpoint contract (`checkpoint['modelG_state_dict']` must load with
strict=True, reference utils/model_save_util.py:188-198).  Conv2d weights are (Cout, Cin/groups, kh, kw);
ConvTranspose2d weights are (Cin, Cout, kh, kw) — transposed layers are flagged.
"""


def generator_spec(filters=32, layer_factor=4, unet_norm="none", bilinear=0, up_mode=0):
    """[(key, shape, kind)], kind in {"conv", "convT", "bias", "buffer", "embed"} in state_dict order.  `layer_factor` = members of
    the skip concatenation (4 for the published square_and_square_root, 2 for original_unet, 3 for square / square_root:
    unet_parts.py:311-332; the first decoder convolution takes layer_factor x C channels).  `bilinear` = 1: the decoder's `up` is
    nn.Sequential(nn.Upsample(scale_factor=2), nn.Conv2d(C, C, 1)) (unet_parts.py:256-259), key `...up.1`; `up_mode` = 1: the
    parameter-free zero-insertion upsampling of unet_parts.py:284-288 (no `up` entries at all).  With
    unet_norm='batch_norm' every 3x3 convolution of the double-conv blocks is followed by its nn.BatchNorm2d entries
    (unet_parts.py:20-21, 34-35: `norm` after `conv`, `norm1` after `conv1`), kinds "bn_weight", "bn_bias", "bn_mean", "bn_var",
    "bn_count"; InstanceNorm2d (no affine, no running statistics) adds nothing."""
    f = filters
    spec = []
    bn = unet_norm == "batch_norm"

    def conv(p, cin, cout, k, transposed=False):
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        spec.append((p + ".weight", shape, "convT" if transposed else "conv"))
        spec.append((p + ".bias", (cout,), "bias"))
        if bn and k == 3:
            q = p[:-len("conv")] + "norm" if p.endswith(".conv") else p[:-len("conv1")] + "norm1"
            spec.append((q + ".weight", (cout,), "bn_weight"))
            spec.append((q + ".bias", (cout,), "bn_bias"))
            spec.append((q + ".running_mean", (cout,), "bn_mean"))
            spec.append((q + ".running_var", (cout,), "bn_var"))
            spec.append((q + ".num_batches_tracked", (), "bn_count"))

    conv("inc.conv.conv", 1, f, 3)
    conv("inc.conv.conv1", f, f, 3)
    ch = f
    for i in range(3):
        conv("down_path.%d.mpconv.1.conv" % i, ch, ch * 2, 3)
        conv("down_path.%d.mpconv.1.conv1" % i, ch * 2, ch * 2, 3)
        ch *= 2
    conv("down_path.3.mpconv.1.conv", ch, ch, 3)
    conv("down_path.3.mpconv.1.conv1", ch, ch, 3, transposed=True)
    spec.append(("gcn.pos_embed", (1, ch, 12, 12), "embed"))
    g = "gcn.module.0."
    spec.append((g + "0.relative_pos", (1, 144, 144), "buffer"))
    conv(g + "0.fc1.0", ch, ch, 1)
    spec.append((g + "0.graph_conv.gconv.nn.0.weight", (2 * ch, 2 * ch // 4, 1, 1), "conv"))
    spec.append((g + "0.graph_conv.gconv.nn.0.bias", (2 * ch,), "bias"))
    conv(g + "0.fc2.0", 2 * ch, ch, 1)
    conv(g + "1.fc1.0", ch, ch, 1)
    conv(g + "1.fc2.0", ch, ch, 1)
    for i in range(4):
        out = ch // 2 if i < 2 else f
        p = "up_path.%d" % i
        if up_mode:
            pass
        elif bilinear:
            conv(p + ".up.1", ch, ch, 1)
        else:
            conv(p + ".up", ch, ch, 2, transposed=True)
        conv(p + ".conv.conv", ch * layer_factor, out, 3, transposed=True)
        conv(p + ".conv.conv1", out, out, 3, transposed=True)
        ch //= 2
    conv("outc.conv", f, 1, 1)
    return spec


def simple_d_spec(dim=16, input_size=256):
    last = ((input_size // 2 - 1) // 2 - 1) ** 2
    return [("model.0.weight", (dim, 1, 4, 4), "conv"), ("model.0.bias", (dim,), "bias"),
            ("model.2.weight", (2 * dim, dim, 4, 4), "conv"), ("model.2.bias", (2 * dim,), "bias"),
            ("model.4.weight", (1, 2 * dim, 1, 1), "conv"), ("model.4.bias", (1,), "bias"),
            ("tail.1.weight", (1, last), "linear")]


def patch_d_spec(ndf=16, n_layers=3):
    spec = [("model.0.weight", (ndf, 1, 4, 4), "conv"), ("model.0.bias", (ndf,), "bias")]
    idx, mult = 2, 1
    for n in range(1, n_layers + 1):
        prev, mult = mult, min(2 ** n, 8)
        spec.append(("model.%d.conv.weight" % idx, (ndf * mult, ndf * prev, 4, 4), "conv"))
        idx += 1
    spec += [("model.%d.weight" % idx, (1, ndf * mult, 4, 4), "conv"), ("model.%d.bias" % idx, (1,), "bias")]
    return spec


def batch_norm_layers(filters=32, layer_factor=4):
    """[(conv key prefix, norm key prefix)] of the eighteen convolutions a BatchNorm2d follows, e.g.
    ("inc.conv.conv", "inc.conv.norm"), ("up_path.3.conv.conv1", "up_path.3.conv.norm1")."""
    out = []
    for key, _, kind in generator_spec(filters, layer_factor, "batch_norm"):
        if kind == "bn_weight":
            q = key[:-len(".weight")]
            out.append((q[:-len("norm")] + "conv" if q.endswith("norm") else q[:-len("norm1")] + "conv1", q))
    return out
