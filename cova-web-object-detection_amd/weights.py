"""state_dict contract of the hot path and a platform-stable seeded weight generator.

The reference builds its parameters from ``torchvision.models.resnet18(pretrained=True)``
truncated to ``children()[:-5]`` (reference models.py:49-51) plus the modules created in
``CoVA.__init__`` (models.py:65-90) and ``GraphAttentionLayer.__init__`` (models.py:156-165).
ImageNet weights are not available offline, so parity work and the benchmark use weights
drawn from a fixed numpy ``RandomState`` stream (legacy MT19937 => identical on every box),
shaped exactly like the reference's 50 ``state_dict`` entries.
"""
from collections import OrderedDict

import numpy as np
import torch

BACKBONE_CHANNELS = 64  # ResNet-18 conv1/layer1 width (models.py:49-51 keeps conv1..layer1)
BACKBONE_STRIDE = 4     # conv1 stride 2 * maxpool stride 2


def _bn_entries(prefix, c):
    return [
        (prefix + "weight", (c,)),
        (prefix + "bias", (c,)),
        (prefix + "running_mean", (c,)),
        (prefix + "running_var", (c,)),
        (prefix + "num_batches_tracked", ()),
    ]


def backbone_channels(backbone="resnet18"):
    """Channels of the truncated ResNet's output: layer1 of ResNet-18 keeps 64, of ResNet-50 expands to 256."""
    if backbone == "resnet18":
        return BACKBONE_CHANNELS
    if backbone == "resnet50":
        return 4 * BACKBONE_CHANNELS
    raise ValueError("backbone must be 'resnet18' (the reference, models.py:49) or 'resnet50' (extension)")


def gat_prefixes(n_heads=1, n_gat_layers=1):
    """state_dict prefixes of the attention heads, [layer][head].  One single-head layer keeps the
    reference's keys (``gat.W_i.weight`` ..., models.py:78-79,161-165); the multi-head / stacked
    extension nests them as ``gat.layers.<l>.heads.<h>.``."""
    if n_heads == 1 and n_gat_layers == 1:
        return [["gat."]]
    return [["gat.layers.%d.heads.%d." % (l, h) for h in range(n_heads)] for l in range(n_gat_layers)]


def state_dict_spec(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384,
                    bbox_hidden_dim=32, n_additional_feat=0, backbone="resnet18", n_heads=1,
                    n_gat_layers=1):
    """Ordered (key, shape) list, identical to ``reference CoVA(...).state_dict()`` for the defaults.

    Extensions (BASELINE.json configs[2], configs[4]; absent from the reference, defaults keep its
    behaviour): ``backbone="resnet50"`` = torchvision resnet50 ``children()[:-5]`` (3 Bottleneck
    blocks, 256 output channels, torchvision's key names); ``n_heads`` / ``n_gat_layers`` = several
    GraphAttentionLayer heads of hidden_dim/n_heads channels each, concatenated, stacked n_gat_layers
    times (layer 0 reads the n_feat own features, later layers the previous layer's hidden_dim)."""
    c = BACKBONE_CHANNELS
    spec = [("convnet.0.weight", (c, 3, 7, 7))]
    spec += _bn_entries("convnet.1.", c)
    if backbone == "resnet18":
        for blk in (0, 1):
            p = "convnet.4.%d." % blk
            spec.append((p + "conv1.weight", (c, c, 3, 3)))
            spec += _bn_entries(p + "bn1.", c)
            spec.append((p + "conv2.weight", (c, c, 3, 3)))
            spec += _bn_entries(p + "bn2.", c)
    else:
        cout = backbone_channels(backbone)
        for blk in (0, 1, 2):
            p = "convnet.4.%d." % blk
            cin = c if blk == 0 else cout
            spec.append((p + "conv1.weight", (c, cin, 1, 1)))
            spec += _bn_entries(p + "bn1.", c)
            spec.append((p + "conv2.weight", (c, c, 3, 3)))
            spec += _bn_entries(p + "bn2.", c)
            spec.append((p + "conv3.weight", (cout, c, 1, 1)))
            spec += _bn_entries(p + "bn3.", cout)
            if blk == 0:
                spec.append((p + "downsample.0.weight", (cout, c, 1, 1)))
                spec += _bn_entries(p + "downsample.1.", cout)
    n_visual = backbone_channels(backbone) * roi_output_size[0] * roi_output_size[1]
    n_feat = n_visual + bbox_hidden_dim + n_additional_feat
    if bbox_hidden_dim > 0:
        spec += [("bbox_feat_encoder.0.weight", (bbox_hidden_dim, 5)),
                 ("bbox_feat_encoder.0.bias", (bbox_hidden_dim,))]
        spec += _bn_entries("bbox_feat_encoder.1.", bbox_hidden_dim)
    if n_additional_feat > 0:
        spec += _bn_entries("bn_additional_feat.", n_additional_feat)
    n_total = n_feat
    if use_context:
        if hidden_dim % n_heads:
            raise ValueError("hidden_dim must be divisible by n_heads")
        dh = hidden_dim // n_heads
        for l, heads in enumerate(gat_prefixes(n_heads, n_gat_layers)):
            fin = n_feat if l == 0 else hidden_dim
            for p in heads:
                spec += [(p + "W_i.weight", (dh, fin)),
                         (p + "W_j.weight", (dh, fin)),
                         (p + "attention_layer.weight", (1, 2 * dh)),
                         (p + "attention_layer.bias", (1,))]
        n_total += hidden_dim
    spec += [("decoder.1.weight", (n_total, n_total)), ("decoder.1.bias", (n_total,))]
    spec += _bn_entries("decoder.2.", n_total)
    spec += [("decoder.5.weight", (n_classes, n_total)), ("decoder.5.bias", (n_classes,))]
    return spec


def seeded_state_dict(seed=123, logit_gain=1.0, **cfg):
    """Deterministic weights for every entry of :func:`state_dict_spec`.

    Conv weights ~ N(0, 2/fan_out) (the scheme torchvision's ResNet uses), Linear
    weights/biases ~ U(+-1/sqrt(fan_in)) (torch's default), BN affine/running stats away
    from their trivial values so that every BN term is exercised.  ``logit_gain`` scales
    the last Linear so that integer-prediction tests have decisive logit margins.
    """
    rs = np.random.RandomState(seed)
    spec = state_dict_spec(**cfg)
    bn_prefixes = {k.rsplit(".", 1)[0] for k, _ in spec if k.endswith("running_mean")}
    sd = OrderedDict()
    for key, shape in spec:
        prefix, leaf = key.rsplit(".", 1)
        is_bn = prefix in bn_prefixes
        if leaf == "num_batches_tracked":
            sd[key] = torch.tensor(0, dtype=torch.long)
            continue
        if leaf == "running_mean":
            v = 0.1 * rs.standard_normal(shape)
        elif leaf == "running_var":
            v = rs.uniform(0.5, 1.5, shape)
        elif is_bn and leaf == "weight":
            v = rs.uniform(0.5, 1.5, shape)
        elif is_bn and leaf == "bias":
            v = 0.1 * rs.standard_normal(shape)
        elif len(shape) == 4:  # conv
            fan_out = shape[0] * shape[2] * shape[3]
            v = rs.standard_normal(shape) * np.sqrt(2.0 / fan_out)
        elif len(shape) == 2:  # linear weight
            bound = 1.0 / np.sqrt(shape[1])
            v = rs.uniform(-bound, bound, shape)
            if key == "decoder.5.weight":
                v = v * logit_gain
        else:  # Linear bias
            v = rs.uniform(-0.05, 0.05, shape)
            if key == "decoder.5.bias":
                v = v * logit_gain
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())
    return sd
