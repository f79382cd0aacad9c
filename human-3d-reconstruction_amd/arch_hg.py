"""Hourglass-104 (BASELINE configs[3]: "Hourglass-104 multi_pose"): the parameter table as data.

The reference names the architecture (`--arch ... | hourglass`, src/lib/opts.py:61-63; experiments/multi_pose_hg_1x.sh)
but ships no source for it (`_build_model` always calls `dla_net`, trains/trainer.py:165), so there is nothing to cite
line by line: the table follows the published CenterNet definition (Zhou et al., "Objects as Points", `large_hourglass.py`:
`exkp(n=5, nstack=2, dims=[256,256,384,384,384,512], modules=[2,2,2,2,2,4])`, itself CornerNet's backbone), with that
implementation's state_dict key names so its checkpoints load unchanged.  PARITY UNPINNED.

    pre     = convolution(7, 3, 128, stride 2) -> residual(3, 128, 256, stride 2)                       (1/4 resolution)
    kps[i]  = kp_module(5, dims, modules): up1 = `curr_mod` residuals; low1 = residual(stride 2) + residuals;
              low2 = the next kp_module (innermost: `next_mod` residuals); low3 = residuals, last one back to curr_dim;
              out = up1 + nearest_upsample_x2(low3)                                    (max-pool replaced by the stride)
    cnvs[i] = convolution(3, 256, 256);  heads[h][i] = convolution(3, 256, 256, with_bn=False) -> Conv2d(256, C, 1)
    between the stacks: inter = relu(inters_[i](inter) + cnvs_[i](cnv)); inter = inters[i](inter)
"""

N = 5
DIMS = (256, 256, 384, 384, 384, 512)
MODULES = (2, 2, 2, 2, 2, 4)
PRE_DIM = 128
CNV_DIM = 256
BN_EPS = 1e-5


def _bn(shapes, p, c):
    shapes[p + ".weight"] = (c,)
    shapes[p + ".bias"] = (c,)
    shapes[p + ".running_mean"] = (c,)
    shapes[p + ".running_var"] = (c,)
    shapes[p + ".num_batches_tracked"] = ()


def residual_has_skip(cin, cout, stride):
    return stride != 1 or cin != cout


def _residual(shapes, p, cin, cout, stride=1):
    shapes[p + ".conv1.weight"] = (cout, cin, 3, 3)
    _bn(shapes, p + ".bn1", cout)
    shapes[p + ".conv2.weight"] = (cout, cout, 3, 3)
    _bn(shapes, p + ".bn2", cout)
    if residual_has_skip(cin, cout, stride):
        shapes[p + ".skip.0.weight"] = (cout, cin, 1, 1)
        _bn(shapes, p + ".skip.1", cout)


def layer_specs(kind, cin, cout, modules):
    """[(cin, cout, stride)] of a residual sequence: 'layer' (make_layer), 'hg' (make_hg_layer: first one stride 2),
    'revr' (make_layer_revr: the LAST one changes the width)."""
    if kind == "layer":
        return [(cin, cout, 1)] + [(cout, cout, 1)] * (modules - 1)
    if kind == "hg":
        return [(cin, cout, 2)] + [(cout, cout, 1)] * (modules - 1)
    return [(cin, cin, 1)] * (modules - 1) + [(cin, cout, 1)]


def _seq(shapes, p, kind, cin, cout, modules):
    for j, (ci, co, s) in enumerate(layer_specs(kind, cin, cout, modules)):
        _residual(shapes, "%s.%d" % (p, j), ci, co, s)


def _kp_module(shapes, p, n, dims, modules):
    curr_mod, next_mod, curr_dim, next_dim = modules[0], modules[1], dims[0], dims[1]
    _seq(shapes, p + ".up1", "layer", curr_dim, curr_dim, curr_mod)
    _seq(shapes, p + ".low1", "hg", curr_dim, next_dim, curr_mod)
    if n > 1:
        _kp_module(shapes, p + ".low2", n - 1, dims[1:], modules[1:])
    else:
        _seq(shapes, p + ".low2", "layer", next_dim, next_dim, next_mod)
    _seq(shapes, p + ".low3", "revr", next_dim, curr_dim, curr_mod)


def state_dict_shapes(heads, nstack=2):
    """{key: shape} of `get_large_hourglass_net(num_layers, heads, head_conv)` (exkp with nstack stacks)."""
    shapes = {}
    shapes["pre.0.conv.weight"] = (PRE_DIM, 3, 7, 7)
    _bn(shapes, "pre.0.bn", PRE_DIM)
    _residual(shapes, "pre.1", PRE_DIM, DIMS[0], 2)
    for i in range(nstack):
        _kp_module(shapes, "kps.%d" % i, N, DIMS, MODULES)
        shapes["cnvs.%d.conv.weight" % i] = (CNV_DIM, DIMS[0], 3, 3)
        _bn(shapes, "cnvs.%d.bn" % i, CNV_DIM)
    for i in range(nstack - 1):
        _residual(shapes, "inters.%d" % i, DIMS[0], DIMS[0], 1)
        shapes["inters_.%d.0.weight" % i] = (DIMS[0], DIMS[0], 1, 1)
        _bn(shapes, "inters_.%d.1" % i, DIMS[0])
        shapes["cnvs_.%d.0.weight" % i] = (DIMS[0], CNV_DIM, 1, 1)
        _bn(shapes, "cnvs_.%d.1" % i, DIMS[0])
    for head, c in heads.items():
        for i in range(nstack):
            shapes["%s.%d.0.conv.weight" % (head, i)] = (DIMS[0], CNV_DIM, 3, 3)
            shapes["%s.%d.0.conv.bias" % (head, i)] = (DIMS[0],)
            shapes["%s.%d.1.weight" % (head, i)] = (c, DIMS[0], 1, 1)
            shapes["%s.%d.1.bias" % (head, i)] = (c,)
    return shapes


def conv_flops(heads, in_h=512, in_w=512, nstack=2):
    """Algorithmic FLOPs per image (2 per MAC, convs only: the SURVEY 8d convention)."""
    total = [0.0]

    def conv(cout, cin, k, h, w):
        total[0] += 2.0 * h * w * cout * cin * k * k

    def residual(ci, co, s, h, w):
        ho, wo = h // s, w // s
        conv(co, ci, 3, ho, wo)
        conv(co, co, 3, ho, wo)
        if residual_has_skip(ci, co, s):
            conv(co, ci, 1, ho, wo)
        return ho, wo

    def seq(kind, ci, co, m, h, w):
        for a, b, s in layer_specs(kind, ci, co, m):
            h, w = residual(a, b, s, h, w)
        return h, w

    def kp(n, dims, modules, h, w):
        seq("layer", dims[0], dims[0], modules[0], h, w)
        hl, wl = seq("hg", dims[0], dims[1], modules[0], h, w)
        if n > 1:
            kp(n - 1, dims[1:], modules[1:], hl, wl)
        else:
            seq("layer", dims[1], dims[1], modules[1], hl, wl)
        seq("revr", dims[1], dims[0], modules[0], hl, wl)

    h, w = in_h // 2, in_w // 2
    conv(PRE_DIM, 3, 7, h, w)
    h, w = residual(PRE_DIM, DIMS[0], 2, h, w)
    for i in range(nstack):
        kp(N, DIMS, MODULES, h, w)
        conv(CNV_DIM, DIMS[0], 3, h, w)
        for c in heads.values():
            conv(DIMS[0], CNV_DIM, 3, h, w)
            conv(c, DIMS[0], 1, h, w)
        if i < nstack - 1:
            conv(DIMS[0], DIMS[0], 1, h, w)
            conv(DIMS[0], CNV_DIM, 1, h, w)
            residual(DIMS[0], DIMS[0], 1, h, w)
    return total[0]
