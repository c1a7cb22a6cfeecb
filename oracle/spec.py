"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Parameter-name/shape specification of the reference models, written down independently of the
reference's nn.Module tree so that it can be rebuilt on a machine that has no /root/reference.

Follows (names only; shapes re-derived):
  * PoseResNet registration order ............ /root/reference/mvn/models/pose_resnet.py:184-234
  * Bottleneck / BasicBlock members ........... pose_resnet.py:26-35, 57-73
  * GlobalAveragePoolingHead .................. pose_resnet.py:140-162
  * V2VModel / EncoderDecorder members ........ /root/reference/mvn/models/v2v.py:7-66, 69-101, 141-162
  * VolumetricTriangulationNet members ........ /root/reference/mvn/models/triangulation.py:233-242

``make_golden.py`` asserts that these specs equal ``state_dict()`` of the real reference modules
(key order, shapes, dtypes) and stores a digest in tests/golden/spec_digest.json.
"""
from collections import OrderedDict

RESNET_SPEC = {  # pose_resnet.py:177-181
    18: ("basic", [2, 2, 2, 2]),
    34: ("basic", [3, 4, 6, 3]),
    50: ("bottleneck", [3, 4, 6, 3]),
    101: ("bottleneck", [3, 4, 23, 3]),
    152: ("bottleneck", [3, 8, 36, 3]),
}


def _bn(spec, name, c):
    spec[name + ".weight"] = ((c,), "bn_gamma")
    spec[name + ".bias"] = ((c,), "bn_beta")
    spec[name + ".running_mean"] = ((c,), "bn_mean")
    spec[name + ".running_var"] = ((c,), "bn_var")
    spec[name + ".num_batches_tracked"] = ((), "bn_count")


def _conv(spec, name, cout, cin, k, nd, bias):
    spec[name + ".weight"] = ((cout, cin) + (k,) * nd, "conv_w")
    if bias:
        spec[name + ".bias"] = ((cout,), "conv_b")


def _gap_head(spec, p, cin, ncls):
    _conv(spec, p + ".features.0", 512, cin, 3, 2, True)
    _bn(spec, p + ".features.1", 512)
    _conv(spec, p + ".features.4", 256, 512, 3, 2, True)
    _bn(spec, p + ".features.5", 256)
    for i, (o, c) in zip((0, 2, 4), ((512, 256), (256, 512), (ncls, 256))):
        spec[p + ".head.%d.weight" % i] = ((o, c), "lin_w")
        spec[p + ".head.%d.bias" % i] = ((o,), "lin_b")


def pose_resnet_spec(num_layers, num_joints, alg_conf=False, vol_conf=False, prefix="", caffe=False):
    kind, blocks = RESNET_SPEC[num_layers]
    if caffe:   # pose_resnet.py:322-324: style 'caffe' swaps in Bottleneck_CAFFE for ANY depth (same member names/shapes as Bottleneck)
        kind = "bottleneck"
    exp = 4 if kind == "bottleneck" else 1
    s = OrderedDict()
    _conv(s, prefix + "conv1", 64, 3, 7, 2, False)
    _bn(s, prefix + "bn1", 64)
    inpl = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), blocks)):
        stride = 1 if li == 0 else 2
        for bi in range(nb):
            p = "%slayer%d.%d" % (prefix, li + 1, bi)
            st = stride if bi == 0 else 1
            if kind == "bottleneck":
                _conv(s, p + ".conv1", planes, inpl, 1, 2, False); _bn(s, p + ".bn1", planes)
                _conv(s, p + ".conv2", planes, planes, 3, 2, False); _bn(s, p + ".bn2", planes)
                _conv(s, p + ".conv3", planes * 4, planes, 1, 2, False); _bn(s, p + ".bn3", planes * 4)
            else:
                _conv(s, p + ".conv1", planes, inpl, 3, 2, False); _bn(s, p + ".bn1", planes)
                _conv(s, p + ".conv2", planes, planes, 3, 2, False); _bn(s, p + ".bn2", planes)
            if bi == 0 and (st != 1 or inpl != planes * exp):
                _conv(s, p + ".downsample.0", planes * exp, inpl, 1, 2, False)
                _bn(s, p + ".downsample.1", planes * exp)
            inpl = planes * exp
    if alg_conf:
        _gap_head(s, prefix + "alg_confidences", 512 * exp, num_joints)
    if vol_conf:
        _gap_head(s, prefix + "vol_confidences", 512 * exp, 32)
    for i in range(3):
        # ConvTranspose2d weight is (Cin, Cout, 4, 4), no bias (pose_resnet.py:278-286, :331)
        s["%sdeconv_layers.%d.weight" % (prefix, 3 * i)] = ((inpl, 256, 4, 4), "deconv_w")
        _bn(s, "%sdeconv_layers.%d" % (prefix, 3 * i + 1), 256)
        inpl = 256
    _conv(s, prefix + "final_layer", num_joints, 256, 1, 2, True)
    return s


def _res3d(s, p, cin, cout):
    _conv(s, p + ".res_branch.0", cout, cin, 3, 3, True); _bn(s, p + ".res_branch.1", cout)
    _conv(s, p + ".res_branch.3", cout, cout, 3, 3, True); _bn(s, p + ".res_branch.4", cout)
    if cin != cout:
        _conv(s, p + ".skip_con.0", cout, cin, 1, 3, True); _bn(s, p + ".skip_con.1", cout)


def _basic3d(s, p, cin, cout, k):
    _conv(s, p + ".block.0", cout, cin, k, 3, True); _bn(s, p + ".block.1", cout)


def _up3d(s, p, cin, cout):
    s[p + ".block.0.weight"] = ((cin, cout, 2, 2, 2), "deconv_w")
    s[p + ".block.0.bias"] = ((cout,), "conv_b")
    _bn(s, p + ".block.1", cout)


# (name, kind, cin, cout) in registration order, v2v.py:73-101
V2V_ENCDEC = [
    ("encoder_res1", "res", 32, 64), ("encoder_res2", "res", 64, 128), ("encoder_res3", "res", 128, 128),
    ("encoder_res4", "res", 128, 128), ("encoder_res5", "res", 128, 128), ("mid_res", "res", 128, 128),
    ("decoder_res5", "res", 128, 128), ("decoder_upsample5", "up", 128, 128),
    ("decoder_res4", "res", 128, 128), ("decoder_upsample4", "up", 128, 128),
    ("decoder_res3", "res", 128, 128), ("decoder_upsample3", "up", 128, 128),
    ("decoder_res2", "res", 128, 128), ("decoder_upsample2", "up", 128, 64),
    ("decoder_res1", "res", 64, 64), ("decoder_upsample1", "up", 64, 32),
    ("skip_res1", "res", 32, 32), ("skip_res2", "res", 64, 64), ("skip_res3", "res", 128, 128),
    ("skip_res4", "res", 128, 128), ("skip_res5", "res", 128, 128),
]


def v2v_spec(cin, cout, prefix=""):
    s = OrderedDict()
    _basic3d(s, prefix + "front_layers.0", cin, 16, 7)
    _res3d(s, prefix + "front_layers.1", 16, 32)
    _res3d(s, prefix + "front_layers.2", 32, 32)
    _res3d(s, prefix + "front_layers.3", 32, 32)
    for name, kind, ci, co in V2V_ENCDEC:
        (_res3d if kind == "res" else _up3d)(s, prefix + "encoder_decoder." + name, ci, co)
    _res3d(s, prefix + "back_layers.0", 32, 32)
    _basic3d(s, prefix + "back_layers.1", 32, 32, 1)
    _basic3d(s, prefix + "back_layers.2", 32, 32, 1)
    _conv(s, prefix + "output_layer", cout, 32, 1, 3, True)
    return s


def vol_net_spec(num_layers=152, num_joints=17, vol_conf=False):
    s = pose_resnet_spec(num_layers, num_joints, False, vol_conf, "backbone.")
    _conv(s, "process_features.0", 32, 256, 1, 2, True)
    s.update(v2v_spec(32, num_joints, "volume_net."))
    return s


def alg_net_spec(num_layers=50, num_joints=17, use_confidences=True):
    return pose_resnet_spec(num_layers, num_joints, use_confidences, False, "backbone.")
