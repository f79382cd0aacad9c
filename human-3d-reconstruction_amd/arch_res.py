"""ResNet-101-DCN (BASELINE configs[4]: `ctdet_coco_resdcn101`, experiments/ctdet_coco_resdcn101.sh:3): the parameter
table as data.

The reference only names the architecture (`--arch resdcn_101`, the experiment script); it has no source for it, so the
table follows the published CenterNet definition (`resnet_dcn.py`: torchvision-style ResNet-101 trunk, three up-sampling
stages of DCN(3x3) + BN + ReLU + ConvTranspose2d(4, stride 2, pad 1, bias=False) + BN + ReLU with 256 / 128 / 64 channels,
heads Conv3x3(64 -> head_conv 64) + ReLU + Conv1x1) with that implementation's state_dict names.  PARITY UNPINNED.
"""

LAYERS = {18: None, 101: (3, 4, 23, 3), 50: (3, 4, 6, 3)}
PLANES = (64, 128, 256, 512)
DECONV = (256, 128, 64)
EXPANSION = 4
BN_EPS = 1e-5


def _bn(shapes, p, c):
    shapes[p + ".weight"] = (c,)
    shapes[p + ".bias"] = (c,)
    shapes[p + ".running_mean"] = (c,)
    shapes[p + ".running_var"] = (c,)
    shapes[p + ".num_batches_tracked"] = ()


def blocks(num_layers=101):
    """[(prefix, inplanes, planes, stride, has_downsample)] of the bottleneck trunk."""
    out = []
    inplanes = 64
    for li, (planes, n) in enumerate(zip(PLANES, LAYERS[num_layers]), start=1):
        for b in range(n):
            stride = 2 if (b == 0 and li > 1) else 1
            down = b == 0 and (stride != 1 or inplanes != planes * EXPANSION)
            out.append(("layer%d.%d" % (li, b), inplanes, planes, stride, down))
            inplanes = planes * EXPANSION
    return out


def state_dict_shapes(heads, head_conv=64, num_layers=101):
    shapes = {"conv1.weight": (64, 3, 7, 7)}
    _bn(shapes, "bn1", 64)
    for p, cin, planes, stride, down in blocks(num_layers):
        shapes[p + ".conv1.weight"] = (planes, cin, 1, 1)
        _bn(shapes, p + ".bn1", planes)
        shapes[p + ".conv2.weight"] = (planes, planes, 3, 3)
        _bn(shapes, p + ".bn2", planes)
        shapes[p + ".conv3.weight"] = (planes * EXPANSION, planes, 1, 1)
        _bn(shapes, p + ".bn3", planes * EXPANSION)
        if down:
            shapes[p + ".downsample.0.weight"] = (planes * EXPANSION, cin, 1, 1)
            _bn(shapes, p + ".downsample.1", planes * EXPANSION)
    cin = PLANES[-1] * EXPANSION
    for i, planes in enumerate(DECONV):
        p = "deconv_layers.%d" % (6 * i)
        shapes[p + ".weight"] = (planes, cin, 3, 3)
        shapes[p + ".bias"] = (planes,)
        shapes[p + ".conv_offset_mask.weight"] = (27, cin, 3, 3)
        shapes[p + ".conv_offset_mask.bias"] = (27,)
        _bn(shapes, "deconv_layers.%d" % (6 * i + 1), planes)
        shapes["deconv_layers.%d.weight" % (6 * i + 3)] = (planes, planes, 4, 4)      # ConvTranspose2d: [in, out, kh, kw]
        _bn(shapes, "deconv_layers.%d" % (6 * i + 4), planes)
        cin = planes
    for head, c in heads.items():
        if head_conv > 0:
            shapes[head + ".0.weight"] = (head_conv, DECONV[-1], 3, 3)
            shapes[head + ".0.bias"] = (head_conv,)
            shapes[head + ".2.weight"] = (c, head_conv, 1, 1)
            shapes[head + ".2.bias"] = (c,)
        else:
            shapes[head + ".weight"] = (c, DECONV[-1], 1, 1)
            shapes[head + ".bias"] = (c,)
    return shapes


def conv_flops(heads, in_h=768, in_w=768, head_conv=64, num_layers=101):
    """Algorithmic FLOPs per image (2 per MAC; convs, DCN main + offset convs, transposed convs: 4 of their 16 taps reach
    an output pixel)."""
    t = 0.0
    h, w = in_h // 2, in_w // 2
    t += 2.0 * h * w * 64 * 147
    h, w = h // 2, w // 2
    for p, cin, planes, stride, down in blocks(num_layers):
        t += 2.0 * h * w * planes * cin                       # conv1 at the input resolution
        h, w = h // stride, w // stride
        t += 2.0 * h * w * planes * planes * 9 + 2.0 * h * w * planes * EXPANSION * planes
        if down:
            t += 2.0 * h * w * planes * EXPANSION * cin
    cin = PLANES[-1] * EXPANSION
    for planes in DECONV:
        t += 2.0 * h * w * (planes + 27) * cin * 9
        h, w = 2 * h, 2 * w
        t += 2.0 * h * w * planes * planes * 4
        cin = planes
    for c in heads.values():
        t += 2.0 * h * w * (head_conv * DECONV[-1] * 9 + c * head_conv) if head_conv > 0 else 2.0 * h * w * c * DECONV[-1]
    return t
