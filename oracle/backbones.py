"""oracle/backbones.py -- TEST INFRASTRUCTURE ONLY (CPU restatement; never on the product path).

Functional (state_dict in, tensor out) restatement of the two backbones of the reference,
on torch CPU f32 ops:

* ``resnet_trunk``  -- models/resnet.py:135-151 with ``include_top=False``
  (Bottleneck.forward :57-76, BasicBlock.forward :16-31, _make_layer :111-133).
* ``hardnet_trunk`` -- models/hardnet.py:198-201 (HarDBlock.forward :99-121,
  get_link :58-75, trunk construction :154-196), depth_wise=True only.

The structure is recovered from the state_dict key names / weight shapes, which are the
reference's checkpoint contract.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

_BN_EPS = 1e-5  # nn.BatchNorm2d default, used everywhere in models/*.py


_CALIBRATE = {"on": False}


def _bn(sd, p, x):
    if _CALIBRATE["on"]:   # test-data generation only: give BN the statistics a trained net would hold
        sd[p + ".running_mean"] = x.mean(dim=(0, 2, 3))
        sd[p + ".running_var"] = x.var(dim=(0, 2, 3), unbiased=False).clamp_min(1e-6)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, _BN_EPS)


# --------------------------------------------------------------------------- ResNet
def _res_block(sd, p, x, stride):
    """One residual block.  Bottleneck if ``conv3`` exists (stride on the 3x3: v1.5,
    models/resnet.py:47-48), else BasicBlock (stride on conv1, :8-9).  One PReLU slope
    tensor per block shared by all its activations (:54 / :11)."""
    a = sd[p + ".relu.weight"]
    identity = x
    if (p + ".downsample.0.weight") in sd:                       # models/resnet.py:114-116
        identity = _bn(sd, p + ".downsample.1",
                       F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
    if (p + ".conv3.weight") in sd:
        w2 = sd[p + ".conv2.weight"]
        groups = w2.shape[0] // w2.shape[1]
        out = F.prelu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])), a)
        out = F.prelu(_bn(sd, p + ".bn2", F.conv2d(out, w2, None, stride, 1, 1, groups)), a)
        out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    else:
        out = F.prelu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)), a)
        out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1))
    return F.prelu(out + identity, a)


def resnet_trunk(sd, x, prefix="", upto=4, return_stages=False):
    """ResNet.forward with include_top=False (models/resnet.py:135-146)."""
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x = F.conv2d(x, g["conv1.weight"], None, 2, 3)
    x = F.prelu(_bn(g, "bn1", x), g["relu.weight"])
    x = F.max_pool2d(x, 3, 2, 1)
    stages = [x]
    for li in range(1, upto + 1):
        bi = 0
        while f"layer{li}.{bi}.conv1.weight" in g:
            stride = 2 if (bi == 0 and li > 1) else 1
            x = _res_block(g, f"layer{li}.{bi}", x, stride)
            bi += 1
        stages.append(x)
    if _CALIBRATE["on"]:
        sd.update({prefix + k: v for k, v in g.items()})
    return (x, stages) if return_stages else x


# --------------------------------------------------------------------------- HarDNet
_HARDNET_ARCH = {  # models/hardnet.py:126-152
    68: dict(first=(32, 64), grmul=1.7, gr=(14, 16, 20, 40, 160), n_layers=(8, 16, 16, 16, 4),
             ch_list=(128, 256, 320, 640, 1024), down=(1, 0, 1, 1, 0)),
    85: dict(first=(48, 96), grmul=1.7, gr=(24, 24, 28, 36, 48, 256), n_layers=(8, 16, 16, 16, 16, 4),
             ch_list=(192, 256, 320, 480, 720, 1024), down=(1, 0, 1, 0, 1, 0)),
    39: dict(first=(24, 48), grmul=1.6, gr=(16, 20, 64, 160), n_layers=(4, 16, 8, 4),
             ch_list=(96, 320, 640, 1024), down=(1, 1, 1, 0)),
}


def hard_links(layer: int):
    """Link list of HarDBlock layer ``layer`` >= 1 (models/hardnet.py:58-75): layer - 2^i for
    every i with layer % 2^i == 0, newest first."""
    out, i = [], 0
    while (1 << i) <= layer and i < 10:
        if layer % (1 << i) == 0:
            out.append(layer - (1 << i))
        i += 1
    return out


def _conv_bn_relu6(sd, p, x, stride=1):
    w = sd[p + ".conv.weight"]
    return F.relu6(_bn(sd, p + ".norm", F.conv2d(x, w, None, stride, w.shape[-1] // 2)))


def _dw_bn(sd, p, x, stride=1):
    w = sd[p + ".dwconv.weight"]
    return _bn(sd, p + ".norm", F.conv2d(x, w, None, stride, 1, 1, w.shape[0]))


def _hard_block(sd, p, x, n_layers):
    """HarDBlock.forward (models/hardnet.py:99-121), keepBase=False, dwconv=True."""
    outs = [x]
    for li in range(1, n_layers + 1):
        srcs = [outs[k] for k in hard_links(li)]
        inp = torch.cat(srcs, 1) if len(srcs) > 1 else srcs[0]
        q = f"{p}.layers.{li - 1}"
        outs.append(_dw_bn(sd, q + ".layer2", _conv_bn_relu6(sd, q + ".layer1", inp)))
    t = len(outs)
    return torch.cat([outs[i] for i in range(t) if i == t - 1 or i % 2 == 1], 1)


def hardnet_trunk(sd, x, arch=39, prefix=""):
    """HarDNetFeatureExtraction(depth_wise=True, arch).forward (models/hardnet.py:154-201).
    Any ``arch`` other than 39/85 selects the HarDNet-68 table, as the reference does."""
    cfg = _HARDNET_ARCH[arch if arch in (39, 85) else 68]
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    n = 0
    x = _conv_bn_relu6(g, f"base.{n}", x, stride=2); n += 1       # 3x3 s2
    x = _conv_bn_relu6(g, f"base.{n}", x); n += 1                 # 1x1
    x = _dw_bn(g, f"base.{n}", x, stride=2); n += 1               # dw3x3 s2
    blks = len(cfg["n_layers"])
    for i in range(blks):
        x = _hard_block(g, f"base.{n}", x, cfg["n_layers"][i]); n += 1
        if i == blks - 1 and arch == 85:
            n += 1                                                # nn.Dropout: identity in eval
        x = _conv_bn_relu6(g, f"base.{n}", x); n += 1
        if cfg["down"][i] == 1:
            x = _dw_bn(g, f"base.{n}", x, stride=1); n += 1       # "downsample" at stride 1 (Q13)
    c = cfg["ch_list"][-1]
    x = F.conv2d(x, g[f"base.{n}.weight"], g[f"base.{n}.bias"], 2, 1, 1, c); n += 1
    x = F.relu(x); n += 1
    x = F.conv2d(x, g[f"base.{n}.weight"], g[f"base.{n}.bias"], 2, 1, 1, c); n += 1
    x = F.conv2d(x, g[f"base.{n}.weight"], g[f"base.{n}.bias"], 1, 0, 1, 512)
    if _CALIBRATE["on"]:
        sd.update({prefix + k: v for k, v in g.items()})
    return x


def calibrate_bn(sd, x, trunk, **kw):
    """Overwrite every BatchNorm's running statistics in ``sd`` with the batch statistics of ``x``
    flowing through ``trunk`` (resnet_trunk / hardnet_trunk).  Synthetic-weight conditioning for
    tests: a random-init HarDNet with identity BN squashes the image signal to a spatially constant
    feature map, which makes ~all RPN scores tie exactly - a case in which the reference's own
    (unstable) argsort has no defined order."""
    _CALIBRATE["on"] = True
    try:
        with torch.no_grad():
            trunk(sd, x, **kw)
    finally:
        _CALIBRATE["on"] = False
    return sd
