"""Drop-in mirror of the reference's ``models.py``: ``CoVA`` and ``GraphAttentionLayer``.

Same constructor signature (reference models.py:10-21, called positionally with 9 arguments at
main.py:122-132 / evaluate.py:187-197), same ``forward`` contract (models.py:94-122), same public
members used by the reference's scripts (``n_classes``, ``class_names``, ``gat(...,
return_attn_wts)``, ``_get_visual_features``, ``_get_bbox_features``, ``bn_additional_feat``;
extract_attn_wts_and_visualize.py:117-124) and the same 50 ``state_dict`` keys, so reference
checkpoints load unchanged and ``train.py`` / ``evaluate.py`` can drive it as they drive the
original.

Everything on the device is executed by the hand-written HIP kernels behind
``include/cova_hip.h``; the ``nn.Conv2d`` / ``nn.BatchNorm*`` / ``nn.Linear`` objects below are
parameter containers only (they give the reference's parameter names and shapes) and their own
``forward`` is never used.  Inputs must live on a ROCm device; there is no CPU fallback.
"""
import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import engine
from .weights import backbone_channels

GAT_MAX_K = engine.GAT_MAX_K

_FIELDS = ("roi_output_size", "n_classes", "use_context", "hidden_dim", "bbox_hidden_dim",
           "n_additional_feat", "drop_prob")


class _ParamBlock(nn.Module):
    """BasicBlock-shaped parameter holder: conv1, bn1, relu, conv2, bn2 (torchvision naming)."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        raise RuntimeError("parameter container only; the conv stack runs in libcova_hip.so")


class _ParamBottleneck(nn.Module):
    """torchvision Bottleneck-shaped parameter holder (resnet50 layer1; extension): conv1 1x1, bn1,
    conv2 3x3, bn2, conv3 1x1, bn3, relu, optional downsample = Sequential(conv 1x1, bn)."""

    def __init__(self, cin, planes, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, 4 * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * planes)
        self.relu = nn.ReLU(inplace=True)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * planes, 1, bias=False),
                                            nn.BatchNorm2d(4 * planes))

    def forward(self, x):
        raise RuntimeError("parameter container only; the conv stack runs in libcova_hip.so")


class _RoIPoolSpec(nn.Module):
    """Holds (output_size, spatial_scale) like torchvision.ops.RoIPool (models.py:58)."""

    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale


def _named_tensors(module):
    params = {k: v for k, v in module.named_parameters()}
    buffers = {k: v for k, v in module.named_buffers()}
    return params, buffers


def _require_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "cova_web_object_detection_amd runs on MI355X only: got a %s tensor.  Move the model "
                "and its inputs to the ROCm device (there is no CPU fallback)." % t.device)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def _i64c(t):
    return t.detach().to(torch.int64).contiguous()


def _verify_batch_size(training, n, width):
    """torch.nn.functional.batch_norm's train-mode check (one value per channel has no variance)."""
    if training and n == 1:
        raise ValueError("Expected more than 1 value per channel when training, got input size "
                         "torch.Size([1, %d])" % width)


def _saved_or_raise(sv):
    """The saved activations are released by the first backward (like autograd's saved tensors)."""
    if sv is None:
        raise RuntimeError("Trying to backward through the graph a second time: the saved activations of "
                           "this CoVA forward have already been freed (the HIP path does not support "
                           "retain_graph).")


# ------------------------------------------------------------------------------------- autograd
class _CoVAFn(torch.autograd.Function):
    """Whole forward pass as one autograd node: backward runs engine.model_bwd."""

    @staticmethod
    def forward(ctx, model, need_grad, images, bboxes, additional_feats, context_indices,
                *param_values):
        keys = model._param_keys
        params = dict(zip(keys, [p.detach() for p in param_values]))
        _, buffers = _named_tensors(model)
        seeds = model._next_dropout_seeds()
        logits, sv = engine.model_fwd(model._cfg, params, buffers, _f32c(images), _f32c(bboxes),
                                      _f32c(additional_feats), _i64c(context_indices),
                                      model.training, seeds, model._forced_masks, save=need_grad)
        ctx.sv, ctx.params, ctx.keys = sv, params, keys
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        _saved_or_raise(ctx.sv)
        want_dimg = bool(ctx.needs_input_grad[2])            # images.requires_grad (the reference gets it from autograd)
        grads = engine.model_bwd(ctx.sv, dlogits.contiguous(), ctx.params, want_dimg=want_dimg)
        ctx.sv = None
        return (None, None, grads.get("__images__"), None, None, None) + tuple(grads.get(k) for k in ctx.keys)


class _VisualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, need_grad, images, bboxes, *param_values):
        keys = model._conv_keys
        params = dict(zip(keys, [p.detach() for p in param_values]))
        _, buffers = _named_tensors(model)
        images, bboxes = _f32c(images), _f32c(bboxes)
        feat, sv = engine.convstack_fwd(images, params, buffers, model.training, need_grad)
        out = torch.empty((bboxes.shape[0], model.n_visual_feat), device=images.device)
        if model._cfg["roi_op"] == "align":
            rsv = engine.roialign_fwd(feat, bboxes, model.roi_pool.output_size, model.roi_pool.spatial_scale,
                                      model._cfg["sampling_ratio"], model._cfg["roi_aligned"], out,
                                      model.n_visual_feat)
        else:
            rsv = engine.roipool_fwd(feat, bboxes, model.roi_pool.output_size,
                                     model.roi_pool.spatial_scale, out, model.n_visual_feat)
        ctx.sv, ctx.rsv, ctx.keys, ctx.nv, ctx.params = sv, rsv, keys, model.n_visual_feat, params
        return out

    @staticmethod
    def backward(ctx, gout):
        bwd = engine.roialign_bwd if ctx.rsv.get("kind") == "align" else engine.roipool_bwd
        gfeat = bwd(ctx.rsv, gout.contiguous(), ctx.nv)
        grads = engine.convstack_bwd(ctx.sv, gfeat, params=ctx.params, want_dimg=bool(ctx.needs_input_grad[2]))
        return (None, None, grads.get("__images__"), None) + tuple(grads.get(k) for k in ctx.keys)


class _BBoxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, bboxes, *param_values):
        keys = model._bbox_keys
        params = dict(zip(keys, [p.detach() for p in param_values]))
        _, buffers = _named_tensors(model)
        bboxes = _f32c(bboxes)
        out = torch.empty((bboxes.shape[0], model.bbox_hidden_dim), device=bboxes.device)
        ctx.sv = engine.bbox_fwd(bboxes, params, buffers, model.training, out, model.bbox_hidden_dim)
        ctx.keys, ctx.hd = keys, model.bbox_hidden_dim
        return out

    @staticmethod
    def backward(ctx, gout):
        grads = engine.bbox_bwd(ctx.sv, gout.contiguous(), ctx.hd)
        return (None, None) + tuple(grads.get(k) for k in ctx.keys)


class _BN1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bn, x, weight, bias):
        x = _f32c(x)
        N, C = x.shape
        params = {"bn.weight": weight.detach(), "bn.bias": bias.detach()}
        buffers = {"bn.running_mean": bn.running_mean, "bn.running_var": bn.running_var,
                   "bn.num_batches_tracked": bn.num_batches_tracked}
        out = torch.empty_like(x)
        ctx.st = engine.bn1d_fwd(x, C, N, C, "bn.", params, buffers, bn.training, out, C, False)
        ctx.x = x
        return out

    @staticmethod
    def backward(ctx, gout):
        x = ctx.x
        N, C = x.shape
        dz = torch.empty_like(x)
        dg, db = engine.bn_backward(gout.contiguous(), C, None, 0, x, C, ctx.st, N, dz, C)
        return None, dz, dg, db


class _HipBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d whose forward/backward run in libcova_hip.so (models.py:73)."""

    def forward(self, x):
        _require_cuda(x)
        if x.dim() != 2 or x.shape[1] != self.num_features:
            raise RuntimeError("expected input [N, %d], got %s" % (self.num_features, tuple(x.shape)))
        if x.shape[0] == 0:
            return x
        _verify_batch_size(self.training, x.shape[0], self.num_features)
        return _BN1dFn.apply(self, x, self.weight, self.bias)


class _GATFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, h_i, context_indices, W_i, W_j, att_w, att_b):
        h = _f32c(h_i)
        N, F = h.shape
        params = {"gat.W_i.weight": W_i.detach(), "gat.W_j.weight": W_j.detach(),
                  "gat.attention_layer.weight": att_w.detach(),
                  "gat.attention_layer.bias": att_b.detach()}
        hp = torch.empty((N, layer.hidden_dim), device=h.device)
        ctx.sv = engine.gat_fwd(h, F, N, F, _i64c(context_indices), params, hp, layer.hidden_dim)
        ctx.params = params
        ctx.mark_non_differentiable(ctx.sv["attn"])
        return hp, ctx.sv["attn"]

    @staticmethod
    def backward(ctx, g, _gattn):
        sv = ctx.sv
        dh = torch.empty((sv["N"], sv["F"]), device=g.device)
        grads = engine.gat_bwd(sv, g.contiguous(), sv["D"], ctx.params, dh, sv["F"], False)
        return (None, dh, None, grads["gat.W_i.weight"], grads["gat.W_j.weight"],
                grads["gat.attention_layer.weight"], grads["gat.attention_layer.bias"])


# ------------------------------------------------------------------------------------- modules
def _check_gat_inputs(h_i, context_indices, in_features):
    """What the reference's layer would reject inside torch (models.py:180-200), before raw pointers go out."""
    if h_i.dim() != 2 or h_i.shape[1] != in_features:
        raise RuntimeError("expected h_i [N, %d] (W_i / W_j input width, models.py:161-162), got %s"
                           % (in_features, tuple(h_i.shape)))
    if context_indices.dim() != 2 or context_indices.shape[0] != h_i.shape[0]:
        raise RuntimeError("expected context_indices [%d, n_context], got %s"
                           % (h_i.shape[0], tuple(context_indices.shape)))
    if context_indices.is_floating_point() or context_indices.dtype == torch.bool:
        raise IndexError("context_indices must be an integer tensor (models.py:186 indexes with it)")
    if context_indices.shape[1] > GAT_MAX_K:
        raise ValueError("n_context > %d (-cs > %d) is not supported by the wave-per-node kernel"
                             % (GAT_MAX_K, GAT_MAX_K // 2))


class GraphAttentionLayer(nn.Module):
    """Single-head additive attention over K padded neighbours (reference models.py:151-212)."""

    def __init__(self, in_features, hidden_dim, alpha=0.2):
        super(GraphAttentionLayer, self).__init__()
        if abs(alpha - engine.LEAKY_SLOPE) > 1e-12:
            raise ValueError("the HIP path is built for the reference's LeakyReLU slope 0.2")
        self.in_features = in_features
        self.hidden_dim = hidden_dim
        self.W_i = nn.Linear(self.in_features, self.hidden_dim, bias=False)
        self.W_j = nn.Linear(self.in_features, self.hidden_dim, bias=False)
        self.attention_layer = nn.Linear(2 * self.hidden_dim, 1)
        self.leakyrelu = nn.LeakyReLU(alpha)

    def forward(self, h_i, context_indices, return_attn_wts=False):
        """h_i [N, in_features]; context_indices int64 [N, n_context] with -1 pads."""
        _require_cuda(h_i, context_indices)
        _check_gat_inputs(h_i, context_indices, self.in_features)
        h_prime, attn = _GATFn.apply(self, h_i, context_indices, self.W_i.weight, self.W_j.weight,
                                     self.attention_layer.weight, self.attention_layer.bias)
        if return_attn_wts:
            return h_prime, attn
        return h_prime


class _GATHeads(nn.Module):
    def __init__(self, in_features, hidden_dim, n_heads):
        super().__init__()
        self.heads = nn.ModuleList([GraphAttentionLayer(in_features, hidden_dim // n_heads)
                                    for _ in range(n_heads)])


class MultiHeadGraphAttention(nn.Module):
    """Extension (BASELINE.json configs[2], [4]; the reference has one single-head layer,
    models.py:78-79): ``n_layers`` stacked layers, each the concatenation of ``n_heads``
    GraphAttentionLayers of hidden_dim/n_heads channels over the same neighbour table.  Same call
    contract as GraphAttentionLayer; the attention weights returned are the last layer's, per head
    [N, n_heads, n_context]."""

    def __init__(self, in_features, hidden_dim, n_heads, n_layers):
        super().__init__()
        if hidden_dim % n_heads:
            raise ValueError("hidden_dim must be divisible by n_heads")
        self.layers = nn.ModuleList([_GATHeads(in_features if l == 0 else hidden_dim, hidden_dim, n_heads)
                                     for l in range(n_layers)])

    def forward(self, h_i, context_indices, return_attn_wts=False):
        h, attn = h_i, None
        for layer in self.layers:
            outs = [head(h, context_indices, True) for head in layer.heads]
            h = torch.cat([o for o, _ in outs], dim=1)
            attn = torch.stack([a for _, a in outs], dim=1)
        return (h, attn) if return_attn_wts else h


class CoVA(nn.Module):
    def __init__(self, roi_output_size, img_H, n_classes, use_context=True, hidden_dim=384,
                 bbox_hidden_dim=32, n_additional_feat=0, drop_prob=0.2, class_names=None,
                 backbone="resnet18", n_heads=1, n_gat_layers=1, backbone_state_dict=None, roi_op="pool",
                 sampling_ratio=2, roi_aligned=False):
        """The first nine arguments exactly as the reference's CoVA (models.py:10-34; called positionally
        at main.py:122-132).  Keyword-only-in-practice extensions, whose defaults are the reference's
        model: ``backbone`` 'resnet18' | 'resnet50' (torchvision ``children()[:-5]`` of either),
        ``n_heads`` / ``n_gat_layers`` (MultiHeadGraphAttention), ``backbone_state_dict`` = a torchvision
        ResNet state_dict (or a path to one) whose conv1 / bn1 / layer1 entries initialise the stack --
        the offline stand-in for the reference's ``pretrained=True`` download (models.py:49).
        ``roi_op='align'`` swaps RoIPool (the reference's operator and the parity path) for torchvision's RoIAlign
        (``sampling_ratio``, ``roi_aligned`` as in ``torchvision.ops.RoIAlign``).
        ``img_H`` is only used for the RoIPool scale (models.py:53-56): pages may be any H x W."""
        if roi_op not in ("pool", "align"):
            raise ValueError("roi_op must be 'pool' (the reference, models.py:58) or 'align'")
        super(CoVA, self).__init__()
        self.n_classes = n_classes
        self.use_context = use_context
        self.hidden_dim = hidden_dim
        self.bbox_hidden_dim = bbox_hidden_dim
        self.n_additional_feat = n_additional_feat
        self.class_names = (np.arange(self.n_classes).astype(str) if class_names is None
                            else class_names)
        roi_output_size = (int(roi_output_size[0]), int(roi_output_size[1]))

        # ---- representation network.  ImageNet weights (models.py:49 pretrained=True) cannot be
        # fetched offline: convs get torchvision's kaiming-normal(fan_out) init unless a torchvision
        # state_dict is supplied (the explicit backbone_state_dict= argument: a dict or a path; no hidden
        # environment state) or a reference checkpoint is loaded afterwards with load_state_dict.
        c = engine.C64
        c_out = backbone_channels(backbone)
        conv1 = nn.Conv2d(3, c, 7, 2, 3, bias=False)
        if backbone == "resnet18":
            layer1 = nn.Sequential(_ParamBlock(c), _ParamBlock(c))
        else:
            layer1 = nn.Sequential(_ParamBottleneck(c, c, True), _ParamBottleneck(c_out, c, False),
                                   _ParamBottleneck(c_out, c, False))
        self.backbone = backbone
        self.convnet = nn.Sequential(conv1, nn.BatchNorm2d(c), nn.ReLU(inplace=True),
                                     nn.MaxPool2d(3, 2, 1), layer1)
        for m in self.convnet.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._init_backbone(backbone_state_dict)
        c = c_out
        # models.py:53-56 reads the output size off a dummy forward; it is a closed form
        feat_h = engine.feature_map_size(img_H)
        self.roi_pool = _RoIPoolSpec(roi_output_size, feat_h / img_H)
        self.n_visual_feat = c * roi_output_size[0] * roi_output_size[1]
        self.n_feat = self.n_visual_feat + self.bbox_hidden_dim + self.n_additional_feat

        if self.bbox_hidden_dim > 0:
            self.bbox_feat_encoder = nn.Sequential(nn.Linear(5, self.bbox_hidden_dim),
                                                   nn.BatchNorm1d(self.bbox_hidden_dim), nn.ReLU())
        if self.n_additional_feat > 0:
            self.bn_additional_feat = _HipBatchNorm1d(self.n_additional_feat)
        else:
            self.bn_additional_feat = lambda x: x

        if self.use_context:
            if n_heads == 1 and n_gat_layers == 1:
                self.gat = GraphAttentionLayer(self.n_feat, self.hidden_dim)
            else:
                self.gat = MultiHeadGraphAttention(self.n_feat, self.hidden_dim, n_heads, n_gat_layers)
        self.n_total_feat = self.n_feat + (self.hidden_dim if self.use_context else 0)
        self.decoder = nn.Sequential(nn.Dropout(drop_prob),
                                     nn.Linear(self.n_total_feat, self.n_total_feat),
                                     nn.BatchNorm1d(self.n_total_feat), nn.ReLU(),
                                     nn.Dropout(drop_prob),
                                     nn.Linear(self.n_total_feat, self.n_classes))

        self._cfg = dict(roi_output_size=roi_output_size, n_classes=n_classes, use_context=use_context,
                         hidden_dim=hidden_dim, bbox_hidden_dim=bbox_hidden_dim,
                         n_additional_feat=n_additional_feat, drop_prob=float(drop_prob),
                         spatial_scale=self.roi_pool.spatial_scale, backbone=backbone,
                         n_heads=n_heads, n_gat_layers=n_gat_layers, roi_op=roi_op,
                         sampling_ratio=int(sampling_ratio), roi_aligned=bool(roi_aligned))
        self._param_keys = [k for k, _ in self.named_parameters()]
        self._conv_keys = [k for k in self._param_keys if k.startswith("convnet.")]
        self._bbox_keys = [k for k in self._param_keys if k.startswith("bbox_feat_encoder.")]
        self._dropout_seed, self._dropout_calls = 0x5EED, 0
        self._forced_masks = None      # parity tests inject keep-masks here
        print("Model Parameters:", sum(p.numel() for p in self.parameters() if p.requires_grad))

    _warned_random_init = set()

    def _init_backbone(self, source):
        """conv1 / bn1 / layer1 of a torchvision ResNet state_dict -> convnet.0 / .1 / .4 (keys map 1:1).
        Without one the stack keeps its random init, which the reference never does: say so."""
        if source is None:
            if self.backbone not in CoVA._warned_random_init:          # once per process and architecture
                CoVA._warned_random_init.add(self.backbone)
                warnings.warn("CoVA: no ImageNet weights for the %s stack (the reference downloads them, "
                              "models.py:49); it starts from a random init.  Pass backbone_state_dict= (a torchvision "
                              "state_dict or a path to one), or load a checkpoint." % self.backbone, stacklevel=3)
            return
        if isinstance(source, (str, bytes, os.PathLike)):
            print("CoVA: backbone weights from", source)
            sd = torch.load(source, map_location="cpu")
        else:
            sd = source
        mapped = {}
        for k, v in sd.items():
            for src, dst in (("conv1.", "0."), ("bn1.", "1."), ("layer1.", "4.")):
                if k.startswith(src):
                    mapped[dst + k[len(src):]] = v
        missing = self.convnet.load_state_dict(mapped, strict=True)
        assert not missing.missing_keys

    # ---------------------------------------------------------------- dropout randomness
    def seed_dropout(self, seed):
        self._dropout_seed, self._dropout_calls = int(seed), 0

    def _next_dropout_seeds(self):
        self._dropout_calls += 1
        base = (self._dropout_seed * 0x9E3779B1 + self._dropout_calls * 2) & 0xFFFFFFFFFFFF
        return base, base + 1

    # ---------------------------------------------------------------- reference surface
    def forward(self, images, bboxes, additional_feats, context_indices):
        """images [B,3,H,W] f32, bboxes [N,5] f32 = [batch_idx,x1,y1,x2,y2], additional_feats
        [N,A] f32, context_indices int64 [N,K] (-1 pads) -> scores [N,n_classes] (models.py:94-122)."""
        _require_cuda(images, bboxes, additional_feats, context_indices)
        engine.check_batch(self._cfg, images, bboxes, additional_feats, context_indices, self.training)
        if bboxes.shape[0] == 0:
            return torch.empty((0, self.n_classes), device=images.device)
        values = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and (images.requires_grad or any(p.requires_grad for p in values))
        return _CoVAFn.apply(self, need_grad, images, bboxes, additional_feats, context_indices,
                             *values)

    def _get_visual_features(self, images, bboxes):
        _require_cuda(images, bboxes)
        if images.dim() != 4 or images.shape[1] != 3 or bboxes.dim() != 2 or bboxes.shape[1] != 5:
            raise RuntimeError("expected images [B, 3, H, W] and bboxes [N, 5], got %s and %s"
                               % (tuple(images.shape), tuple(bboxes.shape)))
        named = dict(self.named_parameters())
        values = [named[k] for k in self._conv_keys]
        need_grad = torch.is_grad_enabled() and (images.requires_grad or any(p.requires_grad for p in values))
        return _VisualFn.apply(self, need_grad, images, bboxes, *values)

    def _get_bbox_features(self, bboxes):
        """[x,y,w,h,asp_ratio] -> Linear -> BN -> ReLU (models.py:129-148)."""
        if self.bbox_hidden_dim > 0:
            _require_cuda(bboxes)
            if bboxes.dim() != 2 or bboxes.shape[1] != 5:
                raise RuntimeError("expected bboxes [N, 5], got %s" % (tuple(bboxes.shape),))
            _verify_batch_size(self.training, bboxes.shape[0], self.bbox_hidden_dim)
            named = dict(self.named_parameters())
            return _BBoxFn.apply(self, bboxes, *[named[k] for k in self._bbox_keys])
        return bboxes[:, :0]
