"""DLA-34 + DLAUp/IDAUp + heads: the parameter table (reference state_dict key names) as data.

The reference builds the network from nn.Module classes (models/model.py:225-292 DLA, 169-222
Tree, 148-166 Root, 346-415 DeformConv/IDAUp/DLAUp, 429-473 DLASeg).  Here the same network is
a table: the engine (engine.py) walks it to emit a launch plan, and model.py hangs parameters
with the reference's state_dict names on it so reference checkpoints load unchanged
(trains/trainer.py:475-509).
"""

LEVELS = (1, 1, 1, 2, 2, 1)              # dla34 (model.py:309-312)
CHANNELS = (16, 32, 64, 128, 256, 512)
BN_EPS = 1e-5


def _bn(shapes, p, c):
    shapes[p + ".weight"] = (c,)
    shapes[p + ".bias"] = (c,)
    shapes[p + ".running_mean"] = (c,)
    shapes[p + ".running_var"] = (c,)
    shapes[p + ".num_batches_tracked"] = ()


def _tree(shapes, p, levels, cin, cout, level_root, root_dim=0):
    root_dim = root_dim or 2 * cout
    if level_root:
        root_dim += cin
    if levels == 1:
        for blk, ci in (("tree1", cin), ("tree2", cout)):
            shapes["%s.%s.conv1.weight" % (p, blk)] = (cout, ci, 3, 3)
            _bn(shapes, "%s.%s.bn1" % (p, blk), cout)
            shapes["%s.%s.conv2.weight" % (p, blk)] = (cout, cout, 3, 3)
            _bn(shapes, "%s.%s.bn2" % (p, blk), cout)
        shapes[p + ".root.conv.weight"] = (cout, root_dim, 1, 1)
        _bn(shapes, p + ".root.bn", cout)
    else:
        _tree(shapes, p + ".tree1", levels - 1, cin, cout, False)
        _tree(shapes, p + ".tree2", levels - 1, cout, cout, False, root_dim + cout)
    if cin != cout:
        shapes[p + ".project.0.weight"] = (cout, cin, 1, 1)
        _bn(shapes, p + ".project.1", cout)


def ida_specs():
    """[(prefix, out_channels, [in channel of proj_k ...], [up factor of up_k ...])]
    DLAUp(startp=2, [64,128,256,512], scales [1,2,4,8]) (model.py:393-407) + ida_up
    (model.py:446-447)."""
    ch = list(CHANNELS[2:])
    in_ch = list(ch)
    scales = [1, 2, 4, 8]
    specs = []
    for i in range(len(ch) - 1):
        j = len(ch) - 2 - i
        specs.append(("dla_up.ida_%d" % i, ch[j], in_ch[j + 1:], [s // scales[j] for s in scales[j + 1:]]))
        for k in range(j + 1, len(ch)):
            scales[k] = scales[j]
            in_ch[k] = ch[j]
    specs.append(("ida_up", ch[0], ch[1:3], [2, 4]))
    return specs


def state_dict_shapes(heads, use_dcn, head_conv=256):
    """{key: shape} for `dla_net(heads, 34, head_conv, 4, not_use_dcn=not use_dcn)`."""
    shapes = {}
    shapes["base.base_layer.0.weight"] = (CHANNELS[0], 3, 7, 7)
    _bn(shapes, "base.base_layer.1", CHANNELS[0])
    shapes["base.level0.0.weight"] = (CHANNELS[0], CHANNELS[0], 3, 3)
    _bn(shapes, "base.level0.1", CHANNELS[0])
    shapes["base.level1.0.weight"] = (CHANNELS[1], CHANNELS[0], 3, 3)
    _bn(shapes, "base.level1.1", CHANNELS[1])
    for lv in range(2, 6):
        _tree(shapes, "base.level%d" % lv, LEVELS[lv], CHANNELS[lv - 1], CHANNELS[lv], lv > 2)
    for prefix, o, ins, ups in ida_specs():
        for k, (ci, f) in enumerate(zip(ins, ups), start=1):
            for name, c_in in (("proj_%d" % k, ci), ("node_%d" % k, o)):
                p = "%s.%s" % (prefix, name)
                _bn(shapes, p + ".actf.0", o)
                shapes[p + ".conv.weight"] = (o, c_in, 3, 3)
                shapes[p + ".conv.bias"] = (o,)
                if use_dcn:
                    shapes[p + ".conv.conv_offset_mask.weight"] = (27, c_in, 3, 3)
                    shapes[p + ".conv.conv_offset_mask.bias"] = (27,)
            shapes["%s.up_%d.weight" % (prefix, k)] = (o, 1, 2 * f, 2 * f)
    for head, c in heads.items():
        if head_conv > 0:
            shapes[head + ".0.weight"] = (head_conv, CHANNELS[2], 3, 3)
            shapes[head + ".0.bias"] = (head_conv,)
            shapes[head + ".2.weight"] = (c, head_conv, 1, 1)
            shapes[head + ".2.bias"] = (c,)
        else:
            shapes[head + ".weight"] = (c, CHANNELS[2], 1, 1)
            shapes[head + ".bias"] = (c,)
    return shapes


def conv_flops(heads, use_dcn, in_h=512, in_w=512, head_conv=256):
    """Algorithmic FLOPs per image (2 FLOP per multiply-accumulate, conv/deconv only -- the
    convention of SURVEY 8d) of the graph the engine actually executes.  The dead
    `project` convs of the two-level trees (level3/level4, model.py:212 overwrites the
    residual) are not executed and not counted."""
    shapes = state_dict_shapes(heads, use_dcn, head_conv)
    res = {"base.base_layer": 1, "base.level0": 1, "base.level1": 2, "base.level2": 4,
           "base.level3": 8, "base.level4": 16, "base.level5": 32}
    total = 0
    for k, s in shapes.items():
        if len(s) != 4:
            continue
        if k in ("base.level3.project.0.weight", "base.level4.project.0.weight"):
            continue
        if k.startswith("base."):
            d = res[".".join(k.split(".")[:2])]
        elif ".up_" in k:
            d = None
        elif k.split(".")[0] in heads:
            d = 4
        else:
            d = None
        if d is None:
            # neck: resolution follows from the IDA spec (proj at the input level's stride, node/up at
            # the output stride)
            d = _neck_stride(k)
        px = (in_h // d) * (in_w // d)
        if ".up_" in k:
            total += 2 * px * s[0] * 4          # 2x2 input taps reach each output pixel
        else:
            total += 2 * px * s[0] * s[1] * s[2] * s[3]
    return total


def _neck_stride(key):
    parts = key.split(".")
    if parts[0] == "dla_up":
        i = int(parts[1].split("_")[1])          # ida_i works at output stride 2^(4-i)... of level (4-i)
        out_stride = 2 ** (4 - i + 0)            # ida_0 -> level4 stride 16, ida_1 -> 8, ida_2 -> 4
        name = parts[2]
    else:
        out_stride = 4
        name = parts[1]
    kind, k = name.split("_")
    k = int(k)
    if kind == "proj":
        if parts[0] == "dla_up":
            return out_stride * 2                # every dla_up proj input is one level coarser
        return out_stride * (2 ** k)             # ida_up: proj_1 @ stride 8, proj_2 @ stride 16
    return out_stride                            # node_k and up_k output
