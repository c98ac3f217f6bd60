"""Stage functions of the hot path: each composes C-ABI calls (libcova_hip.so) on device
buffers owned by PyTorch.  No arithmetic of the path is done by torch operators here --
torch only allocates HBM and provides the stream.

Stages: conv stack R1, RoIPool R2, positional encoder R3, optional
additional-feature BN R4, GAT G1-G6, decoder D, loss/predictions L/P, optimizer U.
``params`` maps the reference's state_dict keys (weights.state_dict_spec) to device tensors.
"""
import ctypes
import os

import torch

from . import _lib
from .weights import BACKBONE_CHANNELS as C64

GAT_MAX_K = 1024       # COVA_GAT_MAX_K of csrc/common.h: neighbour slots per node (up to sixteen 64-lane passes of a wavefront)

call, query = _lib.call, _lib.query
BN_MOMENTUM, BN_EPS, LEAKY_SLOPE = 0.1, 1e-5, 0.2   # nn.BatchNorm defaults; models.py:156


def on_device_of(argpos):
    """Run an engine entry point under the device guard of its tensors (argument `argpos`): the host-side size
    queries (persistent-grid sizes, workspace sizes) then consult the SAME device the launches go to -- the reference
    picks `cuda:<-d>` without ever calling set_device (main.py:17), so the process' current device may be another GPU."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*args, **kw):
            t = args[argpos]
            t = t.get("images") if isinstance(t, dict) else t
            if isinstance(t, torch.Tensor) and t.is_cuda and t.device.index != torch.cuda.current_device():
                with torch.cuda.device(t.device):
                    return fn(*args, **kw)
            return fn(*args, **kw)
        return wrapped
    return deco


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _gbuf(gout, key, shape, like):
    """Gradient destination: the caller's view (flat all-reduce bucket) or a fresh tensor."""
    if gout is not None and key in gout:
        g = gout[key]
        assert g.numel() == int(torch.Size(shape).numel()) and g.is_contiguous(), key
        return g
    return _empty(shape, like)


def _check(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


def check_batch(cfg, images, bboxes, additional_feats, context_indices, training):
    """Shape contract of CoVA.forward (models.py:94-122), checked on the host before any launch (the
    kernels take raw pointers): the errors torch raises inside the reference's forward for malformed
    input -- conv2d's channel check, the [N,5] roi layout, the cat/Linear width mismatch, gather
    broadcasting, BatchNorm1d's train-mode batch-size check -- with the same exception types.
    Index VALUES are not read here (that would cost a device sync): ids outside [0, N) are treated
    as pads by the kernels, memory-safe; the reference raises IndexError for them on CPU."""
    if images.dim() != 4 or images.shape[1] != 3:
        raise RuntimeError("expected images [B, 3, H, W] (conv1 has 3 input channels, models.py:49), got %s"
                           % (tuple(images.shape),))
    if bboxes.dim() != 2 or bboxes.shape[1] != 5:
        raise RuntimeError("expected bboxes [N, 5] = [batch_idx, x1, y1, x2, y2] (models.py:97), got %s"
                           % (tuple(bboxes.shape),))
    N, A = bboxes.shape[0], cfg["n_additional_feat"]
    if additional_feats.dim() != 2 or tuple(additional_feats.shape) != (N, A):
        raise RuntimeError("expected additional_feats [%d, %d] (n_additional_feat, models.py:98,110), got %s"
                           % (N, A, tuple(additional_feats.shape)))
    if cfg["use_context"]:
        if context_indices.dim() != 2 or context_indices.shape[0] != N:
            raise RuntimeError("expected context_indices [%d, 2*context_size] (models.py:99), got %s"
                               % (N, tuple(context_indices.shape)))
        if context_indices.is_floating_point() or context_indices.dtype == torch.bool:
            raise IndexError("context_indices must be an integer tensor (models.py:186 indexes with it)")
        if context_indices.shape[1] > GAT_MAX_K:
            raise ValueError("n_context > %d (-cs > %d) is not supported by the wave-per-node kernel"
                             % (GAT_MAX_K, GAT_MAX_K // 2))
    if training and N == 1:
        # torch.nn.functional.batch_norm's _verify_batch_size, hit by the first BatchNorm1d of the forward
        width = cfg["bbox_hidden_dim"] or A or None
        if width is None:
            width = (backbone_feat(cfg) + (cfg["hidden_dim"] if cfg["use_context"] else 0))
        raise ValueError("Expected more than 1 value per channel when training, got input size "
                         "torch.Size([1, %d])" % width)


def backbone_feat(cfg):
    PH, PW = cfg["roi_output_size"]
    from .weights import backbone_channels
    return backbone_channels(cfg.get("backbone", "resnet18")) * PH * PW


def feature_map_size(n):
    """conv1 (7,2,3) then maxpool (3,2,1): models.py:53-56 does this with a dummy forward."""
    return query("cova_conv_out_size", query("cova_conv_out_size", n, 7, 2, 3), 3, 2, 1)


# ------------------------------------------------------------------------------- BatchNorm
class BNState:
    """scale/shift/mean/invstd of one BatchNorm application (+ what backward needs)."""
    __slots__ = ("scale", "shift", "mean", "invstd", "count", "C", "abc", "frozen")


FOLD_ABOVE, FOLD_GROUP = 256, 64


def fold_partials(partial, nparts, width):
    """Pre-reduce a long list of partial rows so the single-block finalize kernels stay short."""
    while nparts > FOLD_ABOVE:
        n2 = (nparts + FOLD_GROUP - 1) // FOLD_GROUP
        out = _empty((n2, width), partial)
        call("cova_partials_fold", partial, nparts, width, FOLD_GROUP, out)
        partial, nparts = out, n2
    return partial, nparts


# Optional SyncBN (SURVEY.md section 8e "exact large-batch mode"): while STAT_SYNC is set (by the
# trainer, for the duration of one training step) every BatchNorm statistic row -- (sum x, sum x^2)
# forward, (sum dy, sum dy*xhat) backward -- is summed over the data-parallel group before it is
# finalized, with the element count scaled to the whole batch.  Messages are 2*C floats.
class StatSync:
    def __init__(self, group, ratio_pages, ratio_boxes):
        self.group = group
        self.ratio = {"pages": float(ratio_pages), "boxes": float(ratio_boxes)}   # global / local count

    def all_reduce(self, row):
        import torch.distributed as dist
        dist.all_reduce(row, op=dist.ReduceOp.SUM, group=self.group)


STAT_SYNC = None


def _global_stats(partial, nparts, width, count, unit):
    """Local partial rows -> ONE row summed over all ranks, and the count of the whole batch."""
    row = _empty((1, width), partial)
    call("cova_partials_fold", partial, nparts, width, max(int(nparts), 1), row)
    STAT_SYNC.all_reduce(row)
    return row, 1, count * STAT_SYNC.ratio[unit]


def bn_finalize_bwd(part, nparts, C, count, dgamma, dbeta, unit, abc_from=None, frozen=False):
    """cova_bn_finalize_bwd (-> coef [2,C]) or, with ``abc_from`` = the layer's BNState,
    cova_bn_finalize_bwd_abc (-> A|B|C [3,C]).  Under SyncBN the parameter gradients come from the
    LOCAL sums (the gradient exchange adds the ranks up) and the dz coefficients from the sums and
    the count of the whole batch (torch.nn.SyncBatchNorm's backward does the same).
    ``frozen`` (eval-mode BatchNorm, running statistics): the layer is a fixed per-channel affine, so
    dz = scale*dy -- the batch-coupling terms are dropped (A = scale, B = C = 0 / coef = 0) while
    dgamma = sum dy*xhat and dbeta = sum dy keep their meaning (torch's eval-mode batch_norm backward)."""
    part, nparts = fold_partials(part, nparts, 2 * C)
    out = _empty((3 if abc_from is not None else 2, C), part)

    def fin(p, n, cnt, dg, db):
        if abc_from is not None:
            call("cova_bn_finalize_bwd_abc", p, n, C, float(cnt), dg, db, abc_from.mean, abc_from.invstd,
                 abc_from.scale, out)
        else:
            call("cova_bn_finalize_bwd", p, n, C, float(cnt), dg, db, out)

    fin(part, nparts, count, dgamma, dbeta)
    if STAT_SYNC is not None:
        gp, gn, gc = _global_stats(part, nparts, 2 * C, count, unit)
        scratch = _empty((2, C), part)
        fin(gp, gn, gc, scratch[0], scratch[1])
    if frozen:
        out.zero_()
        if abc_from is not None:
            out[0].copy_(abc_from.scale)
    return out


def bn_state(C, like, count, training):
    st = BNState()
    st.C, st.count, st.frozen = C, float(count), not training
    st.abc = _empty((3, C), like)               # scale | (unused) | shift: the prologue's A | B | C
    st.scale, st.shift = st.abc[0], st.abc[2]
    st.mean, st.invstd = _empty((C,), like), _empty((C,), like)
    return st


def tails_on(C=C64):
    """in-launch BatchNorm finalize: 64-channel producers, no SyncBN (a collective sits between partials and finalize)"""
    return OPTIONS.bn_tail and STAT_SYNC is None and C == C64


def bn_tail_fwd(prefix, params, buffers, C, like, count, update_running=True):
    """(BNState, BnTail) for a train-mode BatchNorm whose statistics the NEXT launch produces: the tail writes
    scale / shift / mean / invstd and the running statistics, as cova_bn_finalize_fwd would."""
    st = bn_state(C, like, count, True)
    upd = update_running
    tail = BnTail(1, like, count, gamma=params[prefix + "weight"], beta=params[prefix + "bias"],
                  running_mean=buffers[prefix + "running_mean"] if upd else None,
                  running_var=buffers[prefix + "running_var"] if upd else None,
                  num_batches_tracked=buffers.get(prefix + "num_batches_tracked") if upd else None,
                  scale=st.scale, shift=st.shift, mean=st.mean, invstd=st.invstd)
    return st, tail


def bn_tail_bwd(st, count, gout, prefix, like):
    """(dgamma, dbeta, abc, BnTail): the launch that produces the (sum g, sum g*xhat) partials of BatchNorm `st` also
    writes its parameter gradients and the dz = abc[0]*dy + abc[1]*z + abc[2] coefficients."""
    C = st.C
    dgamma = _gbuf(gout, prefix + "weight", (C,), like)
    dbeta = _gbuf(gout, prefix + "bias", (C,), like)
    abc = _empty((3, C), like)
    tail = BnTail(2, like, count, mean=st.mean, invstd=st.invstd, scale=st.scale, dgamma=dgamma, dbeta=dbeta, abc=abc)
    return dgamma, dbeta, abc, tail


def bn_params(prefix, params, buffers, C, like, training, partial=None, nparts=0, count=0,
              update_running=True, unit="boxes"):
    st = bn_state(C, like, count, training)
    g, b = params[prefix + "weight"], params[prefix + "bias"]
    rm, rv = buffers[prefix + "running_mean"], buffers[prefix + "running_var"]
    if training:
        upd = update_running
        partial, nparts = fold_partials(partial, nparts, 2 * C)
        if STAT_SYNC is not None:
            partial, nparts, count = _global_stats(partial, nparts, 2 * C, count, unit)
            st.count = float(count)
        nbt = buffers.get(prefix + "num_batches_tracked") if upd else None
        call("cova_bn_finalize_fwd", partial, nparts, C, float(count), g, b, rm if upd else None,
             rv if upd else None, nbt, BN_MOMENTUM, BN_EPS, st.scale, st.shift, st.mean, st.invstd)
    else:
        call("cova_bn_eval_params", g, b, rm, rv, BN_EPS, C, st.scale, st.shift, st.mean, st.invstd)
    return st


def colstats(x, ld, R, C):
    n = query("cova_colreduce_num_chunks", R, C)
    part = _empty((n, 2, C), x)
    call("cova_colstats", x, ld, R, C, part)
    return part, n


def bn_backward(dout, ldd, act, lda, z, ldz, st, R, dz, lddz, dres=None, lddres=0, gout=None,
                prefix=None, unit="boxes", drop=None, colsum=None):
    """Returns (dgamma, dbeta); writes dz (and dres = relu-masked dout).  ``drop`` = (mask, p): dout is the gradient
    BEHIND a Dropout of the layer's output (its backward is applied first); ``colsum`` [C]: receives the column sums
    of dz."""
    C = st.C
    if (bn1d_fused(not st.frozen) and dres is None and R > 0 and z.dim() == 2 and dz.data_ptr() != dout.data_ptr()):
        dgamma = _gbuf(gout, (prefix or "") + "weight", (C,), z)
        dbeta = _gbuf(gout, (prefix or "") + "bias", (C,), z)
        call("cova_bn1d_bwd", dout, ldd, drop[0] if drop else None, float(drop[1]) if drop else 0.0, act, lda, z, ldz,
             st.mean, st.invstd, st.scale, R, C, dgamma, dbeta, dz, lddz, colsum)
        return dgamma, dbeta
    if drop is not None:
        dy = _empty((R, C), dout)
        call("cova_dropout_bwd", dout, ldd, drop[0], dy, C, R, C, float(drop[1]))
        dout, ldd = dy, C
    n = query("cova_colreduce_num_chunks", R, C)
    part = _empty((n, 2, C), z)
    call("cova_bn_bwd_reduce", dout, ldd, act, lda, z, ldz, st.mean, st.invstd, R, C, part)
    dgamma = _gbuf(gout, (prefix or "") + "weight", (C,), z)
    dbeta = _gbuf(gout, (prefix or "") + "bias", (C,), z)
    coef = bn_finalize_bwd(part, n, C, R, dgamma, dbeta, unit, frozen=st.frozen)
    call("cova_bn_bwd_apply", dout, ldd, act, lda, z, ldz, st.mean, st.invstd, st.scale, coef, dz,
         lddz, dres, lddres, R, C)
    if colsum is not None:
        call("cova_colsum", dz, lddz, R, C, colsum)
    return dgamma, dbeta


# ------------------------------------------------------------------------------- conv stack
# 3x3 convolutions: Winograd F(4x4,3x3) kernels only (csrc/conv_wino4.hip forward / data gradient, csrc/conv_wgrad4.hip weight
# gradient; the F(2x2,3x3) kernels of rounds 1-3 are a test-support library now, tools/csrc).  BatchNorm+ReLU between
# the two convs of a block and the BatchNorm-backward "apply" passes are evaluated on load inside the consuming
# convolutions: a1 and dz are never written to HBM.
class _BnTailStruct(ctypes.Structure):
    """cova_bn_tail of include/cova_hip.h (a HOST struct of device pointers, read by the launch call)"""
    _fields_ = [("mode", ctypes.c_int), ("counter", ctypes.c_void_p), ("count", ctypes.c_double),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("running_mean", ctypes.c_void_p),
                ("running_var", ctypes.c_void_p), ("num_batches_tracked", ctypes.c_void_p),
                ("momentum", ctypes.c_float), ("eps", ctypes.c_float), ("scale", ctypes.c_void_p),
                ("shift", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("invstd", ctypes.c_void_p),
                ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p), ("abc", ctypes.c_void_p)]


class BnTail:
    """BatchNorm finalize riding on the launch that produces the statistics partials (csrc/bn_tail.h): the struct
    plus the tensors it points to (kept alive until the launch call has returned)."""
    _COUNTERS = {}

    def __init__(self, mode, like, count, **ptrs):
        key = (like.device, torch.cuda.current_stream(like.device).cuda_stream)
        ctr = BnTail._COUNTERS.get(key)
        if ctr is None:                      # one ticket counter per (device, stream): launches on a stream are ordered,
            ctr = BnTail._COUNTERS[key] = torch.zeros((1,), dtype=torch.int32, device=like.device)   # the kernel leaves it 0
        self.keep = [ctr] + [v for v in ptrs.values() if isinstance(v, torch.Tensor)]
        c = _BnTailStruct()
        c.mode, c.counter, c.count = mode, ctr.data_ptr(), float(count)
        c.momentum, c.eps = BN_MOMENTUM, BN_EPS
        for k, v in ptrs.items():
            setattr(c, k, v.data_ptr() if isinstance(v, torch.Tensor) else None)
        self.c = c

    @property
    def ptr(self):
        return ctypes.addressof(self.c)


class Options:
    """The run-time switches of the path (process-wide; read at call time)."""
    # data-parallel steps: the head's gradient all-reduce is issued under the conv-stack backward (trainer.py)
    overlap_allreduce = os.environ.get("COVA_OVERLAP_ALLREDUCE", "1") != "0"
    # BatchNorm finalize as the tail of the producing convolution launch (no separate finalize launches)
    bn_tail = os.environ.get("COVA_BN_TAIL", "1") != "0"


OPTIONS = Options()

CONV3_KEYS = ["convnet.4.0.conv1", "convnet.4.0.conv2", "convnet.4.1.conv1", "convnet.4.1.conv2"]
BN3_KEYS = ["convnet.4.0.bn1.", "convnet.4.0.bn2.", "convnet.4.1.bn1.", "convnet.4.1.bn2."]


class LazyFeature:
    """The last BasicBlock's output relu(bn2(z2) + x) left un-materialised: RoIPool (forward and
    backward mask) forms it on the fly (cova_roipool_fwd_bn / cova_roipool_bwd_bn)."""
    __slots__ = ("z", "x", "scale", "shift", "shape")

    def __init__(self, z, x, scale, shift):
        self.z, self.x, self.scale, self.shift, self.shape = z, x, scale, shift, tuple(z.shape)


def is_bottleneck(params):
    """ResNet-50-stem extension (layer1 = 3 Bottlenecks, 256 channels) vs the reference's ResNet-18."""
    return "convnet.4.0.conv3.weight" in params


@on_device_of(0)
def convstack_fwd(images, params, buffers, training, save=True, lazy_out=False):
    """images NCHW [B,3,H,W] -> feature map NHWC [B,Hf,Wf,C]  (models.py:49-51,125);
    with ``lazy_out`` (Winograd path) a LazyFeature instead of the tensor."""
    _check(images)
    B, _, H, W = images.shape
    H1, W1 = query("cova_conv_out_size", H, 7, 2, 3), query("cova_conv_out_size", W, 7, 2, 3)
    H2, W2 = query("cova_conv_out_size", H1, 3, 2, 1), query("cova_conv_out_size", W1, 3, 2, 1)
    bottleneck = is_bottleneck(params)
    sv = {"images": images, "dims": (B, H, W, H1, W1, H2, W2), "kind": "bottleneck" if bottleneck else "basic"}
    # the Winograd images of the 3x3 weights: one small launch
    if bottleneck:
        sv["_w3"] = conv3_weights([params["convnet.4.%d.conv2.weight" % b] for b in (0, 1, 2)], images)
    else:
        sv["_w3"] = conv3_weights([params[k + ".weight"] for k in CONV3_KEYS], images)
    # conv1 + bn1 + relu + maxpool
    y1 = _empty((B, H1, W1, C64), images)
    nt1 = query("cova_conv1_num_partials", B, H, W)
    part = _empty((nt1, 2, C64), images) if training else None
    # (the kernel reads the OIHW weight itself: no layout-prep launch)
    if training and tails_on():
        bn1, tail = bn_tail_fwd("convnet.1.", params, buffers, C64, images, B * H1 * W1)
        call("cova_conv1_fwd_tail", images, params["convnet.0.weight"], y1, part, B, H, W, tail.ptr)
    else:
        call("cova_conv1_fwd_tail", images, params["convnet.0.weight"], y1, part, B, H, W, None)
        bn1 = bn_params("convnet.1.", params, buffers, C64, images, training, part, nt1, B * H1 * W1, unit="pages")
    p1 = _empty((B, H2, W2, C64), images)
    idx = _empty((B, H2, W2, C64), images, torch.uint8)
    # ymax = y1 at each window's arg-max: lets the last data-gradient conv take bn1's backward sums
    want_ymax = save and (training or bottleneck)
    ymax = _empty((B, H2, W2, C64), images) if want_ymax else None
    call("cova_bn_relu_maxpool_fwd", y1, bn1.scale, bn1.shift, p1, idx, ymax, B, H1, W1)
    sv.update(y1=y1, bn1=bn1, idx=idx, ymax=ymax)
    if bottleneck:
        feat = _layer1_bottleneck_fwd(p1, params, buffers, training, lazy_out, sv)
    else:
        feat = _layer1_basic_fwd(p1, params, buffers, training, save, lazy_out, sv)
    return feat, (sv if save else None)


def conv3_num_partials(B, H, W):
    return query("cova_conv3x3_wino4_num_partials", B, H, W)


def conv3x3_pro(u, inp, in2, abc, relu, addend, act, msc, msh, z, mean, invstd, out, part, B, H, W, tail=None,
                act_bits=None):
    """conv3x3 of f(A*inp + B*in2 + C) (abc / in2 nullable) with the fused epilogue of cova_conv3x3_wino4_full.
    u = the F(4x4,3x3) weight operand (conv3_weights); tail: BnTail; act_bits (with a tail only): the mask source `act`
    as one bit per element (cova_bn_act_fwd_bits)."""
    if tail is not None:
        if act_bits is not None:
            act = None
        call("cova_conv3x3_wino4_full_tail", inp, in2, abc, relu, u, addend, act, act_bits, msc, msh, z, mean, invstd,
             out, part, B, H, W, tail.ptr)
    else:
        call("cova_conv3x3_wino4_full", inp, in2, abc, relu, u, addend, act, msc, msh, z, mean, invstd, out, part,
             B, H, W)


def conv3_weights(ws, like):
    """3x3 weights (a list of up to four OIHW tensors) -> ([forward operands], [data-gradient operands]) of the F(4x4,3x3)
    kernels, all from ONE launch."""
    n = len(ws)
    uf, ud = _empty((n, query("cova_conv3x3_wino4_u_floats")), like), _empty((n, query("cova_conv3x3_wino4_u_floats")), like)
    call("cova_conv3x3_wino4_prep_multi", *(list(ws) + [None] * (4 - n)), uf, ud)
    return [uf[i] for i in range(n)], [ud[i] for i in range(n)]


def conv_bn_fwd(u, inp, abc, relu, out, prefix, params, buffers, training, part, nt, R, B, H, W):
    """3x3 conv (input relu?(abc . inp) on load) followed by a BatchNorm whose statistics it produces -> BNState"""
    if training and tails_on():
        st, tail = bn_tail_fwd(prefix, params, buffers, C64, inp, R)
        conv3x3_pro(u, inp, None, abc, relu, None, None, None, None, None, None, None, out, part, B, H, W, tail)
        return st
    conv3x3_pro(u, inp, None, abc, relu, None, None, None, None, None, None, None, out, part, B, H, W)
    return bn_params(prefix, params, buffers, C64, inp, training, part, nt, R, unit="pages")


def conv_bn_pair_fwd(ua, ub, x, z1, z2, pa, pb, params, buffers, training, part, nt, R, B, H, W):
    """z1 = conv_a(x), bn_a;  z2 = conv_b(relu(bn_a(z1))) with a1 formed on load, bn_b  -> (BNState a, BNState b)"""
    bna = conv_bn_fwd(ua, x, None, 0, z1, pa, params, buffers, training, part, nt, R, B, H, W)
    bnb = conv_bn_fwd(ub, z1, bna.abc, 1, z2, pb, params, buffers, training, part, nt, R, B, H, W)
    return bna, bnb


def _layer1_basic_fwd(p1, params, buffers, training, save, lazy_out, sv):
    """layer1 of ResNet-18: two BasicBlocks (the reference's backbone, models.py:49-51)"""
    images = p1
    B, H, W, H1, W1, H2, W2 = sv["dims"]
    infer = not training and not save
    wf, wd = sv.pop("_w3")                            # (requested in front of the stem, convstack_fwd)
    sv["wd"] = wd
    R = B * H2 * W2
    nt = conv3_num_partials(B, H2, W2)
    x = p1
    blocks = []
    for blk in (0, 1):
        if infer:
            # inference: running statistics are known up front, so BatchNorm (+ residual) + ReLU sit in the conv
            # prologues / epilogues -- two launches per BasicBlock, nothing else touches the maps
            bna = bn_params(BN3_KEYS[2 * blk], params, buffers, C64, images, False)
            bnb = bn_params(BN3_KEYS[2 * blk + 1], params, buffers, C64, images, False)
            ua, ub = wf[2 * blk], wf[2 * blk + 1]
            # conv1 plain; conv2 reads relu(bn1(z1)) formed on load and either carries bn2 + identity + ReLU in its epilogue,
            # or (last block, lazy feature map) leaves them to RoIPool
            z1 = _empty((B, H2, W2, C64), images)
            call("cova_conv3x3_wino4", x, ua, z1, None, B, H2, W2)
            if blk == 1 and lazy_out:
                z2 = _empty((B, H2, W2, C64), images)
                call("cova_conv3x3_wino4_pro", z1, bna.abc, 1, ub, z2, None, B, H2, W2)
                feat = LazyFeature(z2, x, bnb.scale, bnb.shift)
                blocks.append(dict(x=x, z1=z1, a1=None, z2=z2, out=None, bna=bna, bnb=bnb))
                continue
            out = _empty((B, H2, W2, C64), images)
            call("cova_conv3x3_wino4_bnact", z1, bna.abc, 1, ub, x, bnb.scale, bnb.shift, 1, out, B, H2, W2)
            blocks.append(dict(x=x, z1=z1, a1=None, z2=None, out=out, bna=bna, bnb=bnb))
            x = feat = out
            continue
        part = _empty((nt, 2, C64), images) if training else None
        z1, z2 = _empty((B, H2, W2, C64), images), _empty((B, H2, W2, C64), images)
        bna, bnb = conv_bn_pair_fwd(wf[2 * blk], wf[2 * blk + 1], x, z1, z2, BN3_KEYS[2 * blk], BN3_KEYS[2 * blk + 1],
                                    params, buffers, training, part, nt, R, B, H2, W2)
        out_bits = None
        if blk == 1 and lazy_out:
            out = None
            feat = LazyFeature(z2, x, bnb.scale, bnb.shift)
        else:
            out = _empty((B, H2, W2, C64), images)
            if blk == 0 and training and tails_on():
                # ... with its ReLU decisions as bits: the mask source of the next block's conv1 data gradient
                out_bits = _empty((R, 2), images, torch.int32)
                call("cova_bn_act_fwd_bits", z2, bnb.scale, bnb.shift, x, out, out_bits, R)
            else:
                call("cova_bn_act_fwd", z2, C64, bnb.scale, bnb.shift, x, C64, out, C64, R, C64, 1)
            feat = out
        blocks.append(dict(x=x, z1=z1, a1=None, z2=z2, out=out, bna=bna, bnb=bnb, out_bits=out_bits))
        x = out
    sv["blocks"] = blocks
    # what RoIPool's backward needs of the block that produced the feature map
    last = blocks[1]
    sv["last"] = dict(out=last["out"], x=last["x"], z=last["z2"], bn=last["bnb"], prefix=BN3_KEYS[3])
    return feat


# ---- ResNet-50-stem extension: layer1 = 3 torchvision Bottlenecks (conv1x1 Cin->64, bn, relu, conv3x3
# 64->64, bn, relu, conv1x1 64->256, bn, (+ downsample conv1x1 64->256 + bn in block 0), add, relu).
# Same structure as the BasicBlock path: every BatchNorm(+ReLU) between convs is applied on load by the
# consumer, statistics come from the producers' epilogues, the last block's output stays un-materialised.
C256 = 4 * C64


def conv1x1(inp, in2, abc, relu, w, w_trans, out, part, R, cin, cout, addend=None, act=None, msc=None,
            msh=None, z=None, mean=None, invstd=None, z2=None, mean2=None, invstd2=None, part2=None, act_bits=None):
    call("cova_conv1x1", inp, in2, abc, 1 if relu else 0, w, 1 if w_trans else 0, addend, act, act_bits, msc, msh,
         z, mean, invstd, z2, mean2, invstd2, out, part, part2, R, cin, cout)


def _layer1_bottleneck_fwd(p1, params, buffers, training, lazy_out, sv):
    B, H, W, H1, W1, H2, W2 = sv["dims"]
    R = B * H2 * W2
    nt = conv3_num_partials(B, H2, W2)

    def stats(cin, cout):
        n = query("cova_conv1x1_num_partials", R, cin, cout)
        return (_empty((n, 2, cout), p1) if training else None), n

    def bn(prefix, C, part, n):
        return bn_params(prefix, params, buffers, C, p1, training, part, n, R, unit="pages")

    ufs, uds = sv.pop("_w3")
    x, cin, blocks, feat = p1, C64, [], None
    pending = None      # (z3, other, abc): the previous block's output relu(abc . (z3, other)), not yet written
    for blk in (0, 1, 2):
        pre = "convnet.4.%d." % blk
        s = dict(cin=cin, pre=pre)
        part, n = stats(cin, C64)
        s["z1"] = _empty((B, H2, W2, C64), p1)
        if pending is None:
            conv1x1(x, None, None, 0, params[pre + "conv1.weight"], 0, s["z1"], part, R, cin, C64)
        else:
            # the block input is materialised by its first consumer (one pass over the 256-channel map less
            # than a separate bn + residual + ReLU kernel)
            x = _empty((B, H2, W2, C256), p1)
            # ... together with its ReLU decisions, one bit per element: the mask source of this block's input gradient
            s["x_bits"] = _empty((R, C256 // 32), p1, torch.int32) if training else None
            call("cova_conv1x1_materialize", pending[0], pending[1], pending[2], params[pre + "conv1.weight"], x,
                 s["x_bits"], s["z1"], part, R)
            blocks[-1]["out"] = x
        s["x"] = x
        s["bn1"] = bn(pre + "bn1.", C64, part, n)
        uf, s["ud"] = ufs[blk], uds[blk]
        part = _empty((nt, 2, C64), p1) if training else None
        s["z2"] = _empty((B, H2, W2, C64), p1)
        s["bn2"] = conv_bn_fwd(uf, s["z1"], s["bn1"].abc, 1, s["z2"], pre + "bn2.", params, buffers, training, part, nt,
                               R, B, H2, W2)
        part, n = stats(C64, C256)
        s["z3"] = _empty((B, H2, W2, C256), p1)
        conv1x1(s["z2"], None, s["bn2"].abc, 1, params[pre + "conv3.weight"], 0, s["z3"], part, R, C64, C256)
        s["bn3"] = bn(pre + "bn3.", C256, part, n)
        bn3 = s["bn3"]
        s["out"] = None
        if blk == 0:
            part, n = stats(C64, C256)
            s["zd"] = _empty((B, H2, W2, C256), p1)
            conv1x1(x, None, None, 0, params[pre + "downsample.0.weight"], 0, s["zd"], part, R, C64, C256)
            s["bnd"] = bn(pre + "downsample.1.", C256, part, n)
            abc = _empty((3, C256), p1)          # out = relu(s3*z3 + sd*zd + (h3 + hd))
            abc[0].copy_(bn3.scale)
            abc[1].copy_(s["bnd"].scale)
            torch.add(bn3.shift, s["bnd"].shift, out=abc[2])
            pending = (s["z3"], s["zd"], abc)
        elif blk == 1:
            bn3.abc[1].fill_(1.0)                # out = relu(s3*z3 + 1*x + h3): the unused B row of scale | . | shift
            pending = (s["z3"], x, bn3.abc)
        elif lazy_out:
            feat = LazyFeature(s["z3"], x, bn3.scale, bn3.shift)
        else:
            s["out"] = _empty((B, H2, W2, C256), p1)
            call("cova_bn_act_fwd", s["z3"], C256, bn3.scale, bn3.shift, x, C256, s["out"], C256, R, C256, 1)
            feat = s["out"]
        blocks.append(s)
        cin = C256
    sv["blocks"] = blocks
    last = blocks[2]
    sv["last"] = dict(out=last["out"], x=last["x"], z=last["z3"], bn=last["bn3"])
    return feat


def bn_act(z, st, res=None, relu=True):
    """relu?(bn(z) (+ res)) materialised with the same fma the fused prologues evaluate (tests, tools)."""
    C = st.C
    out = torch.empty_like(z)
    call("cova_bn_act_fwd", z, C, st.scale, st.shift, res, C if res is not None else 0, out, C,
         z.numel() // C, C, 1 if relu else 0)
    return out


_IDENT_ABC = {}


def _ident_abc(like):
    """A | B | C = 1 | 0 | 0 for 64 channels (an operand that needs no affine prologue), cached per device."""
    t = _IDENT_ABC.get(like.device)
    if t is None:
        t = torch.zeros((3, C64), device=like.device)
        t[0].fill_(1.0)
        _IDENT_ABC[like.device] = t
    return t


def _lin_conv_bn_bwd(v, act, act_abc, act_relu, w, st, R, gout, bn_prefix, w_key, grads, ws):
    """Backward of (1x1 conv 64->256, BatchNorm ``st``) in linear form (csrc/conv1x1_lin.hip): parameter
    gradients of both, and (avec, m, cvec) = the operands of cova_conv1x1_lin_dgrad.  ``v`` = masked gradient
    w.r.t. the BatchNorm output, ``act`` (+ affine/ReLU on load) = the conv's input.  The conv output is not read."""
    lin = _empty((query("cova_conv1x1_lin_floats"),), v)
    call("cova_conv1x1_vprod", v, act, act_abc, 1 if act_relu else 0, lin, ws, R)
    part = _empty((1, 2, C256), v)
    call("cova_conv1x1_lin_bnsums", lin, w, st.mean, st.invstd, part)
    dg, db, abc = _bn_abc_from_partials(part, 1, st, R, gout, bn_prefix, v)
    grads[bn_prefix + "weight"], grads[bn_prefix + "bias"] = dg, db
    dw = _gbuf(gout, w_key, (C256, C64, 1, 1), v)
    m, cvec, avec = _empty((C64, C64), v), _empty((C64,), v), _empty((3, C256), v)
    call("cova_conv1x1_lin_finish", lin, abc, w, dw, m, cvec, avec)
    grads[w_key] = dw
    return avec, m, cvec


def _layer1_bottleneck_bwd(sv, g, params, gout, grads, head_part):
    """Backward of the three Bottlenecks.  ``g`` = gradient w.r.t. the last block's output, already
    ReLU-masked.  The two 64->256 convolutions (conv3, downsample) and their BatchNorms run in linear form:
    sums, weight gradient and data gradient come from v^T a, a^T a, sum a, sum v, so no 256-channel conv
    output is read in the backward (``head_part``, RoIPool's own sums for the last bn3, is not needed).
    Returns the ReLU-masked gradient w.r.t. the max-pool output (sv['pool_part'] holds the stem's sums)."""
    B, H, W, H1, W1, H2, W2 = sv["dims"]
    R = B * H2 * W2
    nt = conv3_num_partials(B, H2, W2)
    ws3 = _empty((query("cova_conv3x3_wgrad4_workspace_floats", B, H2, W2),), g)
    ws1 = _empty((max(query("cova_conv1x1_wgrad_workspace_floats", R, C256, C64),
                      query("cova_conv1x1_vprod_workspace_floats", R)),), g)
    nd = query("cova_conv1x1_lin_dgrad_num_partials", R)
    stem = sv["bn1"]
    for blk in (2, 1, 0):
        s = sv["blocks"][blk]
        pre, cin, bn1, bn2, bn3 = s["pre"], s["cin"], s["bn1"], s["bn2"], s["bn3"]
        w1, w3 = params[pre + "conv1.weight"], params[pre + "conv3.weight"]
        # conv3 (64->256) + bn3; its input a2 = relu(bn2(z2)) on load, bn2's ReLU mask + sums in the epilogue
        avec, m3, cvec = _lin_conv_bn_bwd(g, s["z2"], bn2.abc, 1, w3, bn3, R, gout, pre + "bn3.",
                                          pre + "conv3.weight", grads, ws1)
        part = _empty((nd, 2, C64), g)
        dy2 = _empty((B, H2, W2, C64), g)
        call("cova_conv1x1_lin_dgrad", g, avec, w3, s["z2"], bn2.abc, 1, m3, cvec, None, bn2.scale, bn2.shift,
             s["z2"], bn2.mean, bn2.invstd, dy2, part, R)
        dg, db, abc2 = _bn_abc_from_partials(part, nd, bn2, R, gout, pre + "bn2.", g)
        grads[pre + "bn2.weight"], grads[pre + "bn2.bias"] = dg, db
        # conv2 (3x3): Winograd weight / data gradient, bn1's ReLU mask + sums in the epilogue
        dw = _gbuf(gout, pre + "conv2.weight", (C64, C64, 3, 3), g)
        # ... with dz2 = abc2 . (dy2, z2) as its side output: the data gradient below reads one tensor, no prologue
        dzm = _empty((B, H2, W2, C64), g)
        call("cova_conv3x3_wgrad4_partial", s["z1"], bn1.abc, 1, dy2, s["z2"], abc2, dzm, ws3, B, H2, W2)
        call("cova_conv3x3_wgrad4_finish", ws3, dw, None, None, None, None, None, None, B, H2, W2)
        d_in, d_in2, d_abc = dzm, None, None
        grads[pre + "conv2.weight"] = dw
        dy1 = _empty((B, H2, W2, C64), g)
        part = _empty((nt, 2, C64), g)
        dg, db, abc1 = dgrad_bn_bwd(s["ud"], d_in, d_in2, d_abc, None, None, bn1.scale, bn1.shift, s["z1"], bn1, dy1, part,
                                    nt, R, gout, pre + "bn1.", B, H2, W2)
        grads[pre + "bn1.weight"], grads[pre + "bn1.bias"] = dg, db
        # conv1 (Cin->64): dz1 = abc1 . (dy1, z1) on load
        dw = _gbuf(gout, pre + "conv1.weight", (C64, cin, 1, 1), g)
        call("cova_conv1x1_wgrad", dy1, s["z1"], abc1, s["x"], None, 0, dw, ws1, R, C64, cin)
        grads[pre + "conv1.weight"] = dw
        if blk > 0:
            # gradient w.r.t. the previous block's output = conv1's data gradient + the identity branch,
            # masked by that output's ReLU (its BatchNorm sums are taken by the next iteration's v^T a)
            dx = _empty((B, H2, W2, C256), g)
            if s.get("x_bits") is not None:
                conv1x1(dy1, s["z1"], abc1, 0, w1, 1, dx, None, R, C64, C256, addend=g, act_bits=s["x_bits"])
            else:
                conv1x1(dy1, s["z1"], abc1, 0, w1, 1, dx, None, R, C64, C256, addend=g, act=s["x"])
            g = dx
        else:
            # downsample branch (64->256 conv + BatchNorm on the block input p1), linear form as well; then
            # dp1 = conv1's + the downsample conv's data gradients, the stem's ReLU mask (from the pooled
            # arg-max pre-activation) and BatchNorm sums
            wdn = params[pre + "downsample.0.weight"]
            avd, md, cvd = _lin_conv_bn_bwd(g, s["x"], None, 0, wdn, s["bnd"], R, gout, pre + "downsample.1.",
                                            pre + "downsample.0.weight", grads, ws1)
            t = _empty((B, H2, W2, C64), g)
            conv1x1(dy1, s["z1"], abc1, 0, w1, 1, t, None, R, C64, C64)
            sv["pool_part"], sv["pool_npart"] = _empty((nd, 2, C64), g), nd
            dp = _empty((B, H2, W2, C64), g)
            call("cova_conv1x1_lin_dgrad", g, avd, wdn, s["x"], _ident_abc(g), 0, md, cvd, t, stem.scale,
                 stem.shift, sv["ymax"], stem.mean, stem.invstd, dp, sv["pool_part"], R)
            g = dp
    return g


def block_a1(blk):
    """a1 = relu(bn1(z1)) of a BasicBlock; materialised on demand when the fused path skipped it
    (same fma expression as the conv prologue, so ReLU decisions are identical)."""
    if blk["a1"] is not None:
        return blk["a1"]
    z1, bna = blk["z1"], blk["bna"]
    a1 = torch.empty_like(z1)
    call("cova_bn_act_fwd", z1, C64, bna.scale, bna.shift, None, 0, a1, C64, z1.numel() // C64, C64, 1)
    return a1


def block_out(blk):
    """out = relu(bn2(z2) + x) of a BasicBlock; materialised on demand when the forward left it lazy."""
    if blk["out"] is not None:
        return blk["out"]
    z2, bnb = blk["z2"], blk["bnb"]
    out = torch.empty_like(z2)
    call("cova_bn_act_fwd", z2, C64, bnb.scale, bnb.shift, blk["x"], C64, out, C64, z2.numel() // C64, C64, 1)
    return out


def bn_bwd_from_partials(part, nparts, dy, z, st, R, dz, gout, prefix):
    """Finalize + apply when the (sum dy, sum dy*xhat) partials were produced by a fused conv
    epilogue and ``dy`` is already ReLU-masked."""
    C = st.C
    dgamma = _gbuf(gout, prefix + "weight", (C,), z)
    dbeta = _gbuf(gout, prefix + "bias", (C,), z)
    coef = bn_finalize_bwd(part, nparts, C, R, dgamma, dbeta, "pages", frozen=st.frozen)
    call("cova_bn_bwd_apply", dy, C, None, 0, z, C, st.mean, st.invstd, st.scale, coef, dz, C, None, 0,
         R, C)
    return dgamma, dbeta


def dgrad_bn_bwd(u, g_in, g_in2, g_abc, addend, act, msc, msh, z, st, out, part, nt, count, gout, prefix, B, H, W,
                 act_bits=None):
    """Data-gradient conv3x3 (input g_abc . (g_in, g_in2) on load, + addend) whose epilogue masks with the ReLU of
    BatchNorm `st` and takes its backward sums; -> (dgamma, dbeta, abc) of that BatchNorm (finalized by the launch's tail
    or by a separate cova_bn_finalize_bwd_abc).  ``count`` = elements per channel of that BatchNorm."""
    if tails_on() and not st.frozen:
        dg, db, abc, tail = bn_tail_bwd(st, count, gout, prefix, out)
        conv3x3_pro(u, g_in, g_in2, g_abc, 0, addend, act, msc, msh, z, st.mean, st.invstd, out, part, B, H, W, tail,
                    act_bits=act_bits)
        return dg, db, abc
    conv3x3_pro(u, g_in, g_in2, g_abc, 0, addend, act, msc, msh, z, st.mean, st.invstd, out, part, B, H, W)
    C = st.C
    dg = _gbuf(gout, prefix + "weight", (C,), out)
    db = _gbuf(gout, prefix + "bias", (C,), out)
    abc = bn_finalize_bwd(part, nt, C, count, dg, db, "pages", abc_from=st, frozen=st.frozen)
    return dg, db, abc


def _bn_abc_from_partials(part, nparts, st, R, gout, prefix, like):
    """(dgamma, dbeta, abc) with dz = abc[0]*dy + abc[1]*z + abc[2] (applied on load downstream)."""
    C = st.C
    dgamma = _gbuf(gout, prefix + "weight", (C,), like)
    dbeta = _gbuf(gout, prefix + "bias", (C,), like)
    abc = bn_finalize_bwd(part, nparts, C, R, dgamma, dbeta, "pages", abc_from=st, frozen=st.frozen)
    return dgamma, dbeta, abc


def _layer1_bwd_fused(sv, dfeat, gout, grads, head_part=None):
    """Backward of the two BasicBlocks with every BatchNorm-backward apply (except the one fed by
    RoIPool's scatter) and both a1 = relu(bn1(z1)) recomputations folded into the Winograd kernels.
    Returns the gradient w.r.t. the max-pool output; sv['pool_abc'] = (dgamma, dbeta, abc) of the stem's BatchNorm."""
    B, H, W, H1, W1, H2, W2 = sv["dims"]
    R = B * H2 * W2
    # weight gradients: the four launches leave their per-block partial sums in four workspaces, ONE launch at the end
    # folds and transforms them (4 x 67 MB at configs[1]; a shared workspace would need a finish launch per convolution)
    wg = "cova_conv3x3_wgrad4"
    nws = query("cova_conv3x3_wgrad4_workspace_floats", B, H2, W2)
    ws_all = _empty((4, nws), dfeat)
    jobs = []

    def wgrad(act, act_abc, act_relu, dz, dz2, dz_abc, dw):
        """-> the gradient operand dz_abc . (dz, dz2) as a materialised map (F(4x4) kernel's side output) or None"""
        out = torch.empty_like(dz) if dz_abc is not None else None
        call(wg + "_partial", act, act_abc, act_relu, dz, dz2, dz_abc, out, ws_all[len(jobs)], B, H2, W2)
        jobs.append(dw)
        return out

    nt = conv3_num_partials(B, H2, W2)
    # head_part = (partials, count): dfeat is already ReLU-masked and the BatchNorm-backward sums of
    # the last bn2 were taken by cova_roipool_bwd_bn
    dA, pend = dfeat, None                  # pend = (dgamma, dbeta, abc) of the bn2 in front of dA, when already known
    if isinstance(head_part, dict):
        pend = head_part["pend"]
    elif head_part is not None:
        last = sv["blocks"][1]["bnb"]
        pend = _bn_abc_from_partials(head_part[0], head_part[1], last, R, gout, BN3_KEYS[3], dfeat)
    for blk in (1, 0):
        s = sv["blocks"][blk]
        ka, kb = CONV3_KEYS[2 * blk], CONV3_KEYS[2 * blk + 1]
        pa, pb = BN3_KEYS[2 * blk], BN3_KEYS[2 * blk + 1]
        bna, bnb = s["bna"], s["bnb"]
        # ---- out = relu(bn2(z2) + x): dz2 = abc_b . (dres, z2); dres = masked incoming gradient
        if pend is None:
            dres, dz2 = torch.empty_like(dA), torch.empty_like(dA)
            dg, db = bn_backward(dA, C64, block_out(s), C64, s["z2"], C64, bnb, R, dz2, C64, dres, C64,
                                 gout, pb, unit="pages")
            g_in, g_in2, g_abc = dz2, None, None
        else:
            dres = dA
            dg, db, g_abc = pend
            g_in, g_in2 = dA, s["z2"]
        grads[pb + "weight"], grads[pb + "bias"] = dg, db
        dw = _gbuf(gout, kb + ".weight", (64, 64, 3, 3), dfeat)
        dzm = wgrad(s["z1"], bna.abc, 1, g_in, g_in2, g_abc, dw)
        grads[kb + ".weight"] = dw
        if dzm is not None:             # the data gradient below reads the materialised operand: one tensor, no prologue
            g_in, g_in2, g_abc = dzm, None, None
        # ---- dgrad of conv2 with bn1's ReLU mask (recomputed from z1) + backward sums in the epilogue
        dy_a = torch.empty_like(dA)
        part = _empty((nt, 2, C64), dfeat)
        dg, db, abc_a = dgrad_bn_bwd(sv["wd"][2 * blk + 1], g_in, g_in2, g_abc, None, None, bna.scale, bna.shift, s["z1"],
                                     bna, dy_a, part, nt, R, gout, pa, B, H2, W2)
        grads[pa + "weight"], grads[pa + "bias"] = dg, db
        dw = _gbuf(gout, ka + ".weight", (64, 64, 3, 3), dfeat)
        dzm = wgrad(s["x"], None, 0, dy_a, s["z1"], abc_a, dw)
        grads[ka + ".weight"] = dw
        d_in, d_in2, d_abc = (dy_a, s["z1"], abc_a) if dzm is None else (dzm, None, None)
        # ---- dgrad of conv1 (+ residual gradient); for block 1 the epilogue prepares block 0's bn2
        dx = torch.empty_like(dA)
        if blk == 1:
            prev = sv["blocks"][0]
            pend = dgrad_bn_bwd(sv["wd"][2 * blk], d_in, d_in2, d_abc, dres, prev["out"], None, None, prev["z2"],
                                prev["bnb"], dx, _empty((nt, 2, C64), dfeat), nt, R, gout, BN3_KEYS[1], B, H2, W2,
                                act_bits=prev.get("out_bits"))
        elif sv.get("ymax") is not None:
            # stem: ReLU mask of bn1 (from the pooled arg-max value) + its backward sums in the epilogue
            bn1 = sv["bn1"]
            sv["pool_abc"] = dgrad_bn_bwd(sv["wd"][2 * blk], d_in, d_in2, d_abc, dres, None, bn1.scale, bn1.shift,
                                          sv["ymax"], bn1, dx, _empty((nt, 2, C64), dfeat), nt, B * H1 * W1, gout,
                                          "convnet.1.", B, H2, W2)
        else:
            conv3x3_pro(sv["wd"][2 * blk], d_in, d_in2, d_abc, 0, dres, None, None, None, None, None, None, dx,
                        None, B, H2, W2)
        dA = dx
    fin = []
    for i in range(4):
        fin += [ws_all[i], jobs[i]]
    call(wg + "_finish", *fin, B, H2, W2)
    return dA


def masked_grad_and_sums(dout, last, R):
    """Stand-alone form of what cova_roipool_bwd_bn fuses: g = dout * (out > 0) and the partial rows of
    (sum g, sum g*xhat(z)) for the BatchNorm in front of the feature map."""
    st, z = last["bn"], last["z"]
    C = st.C
    out = last["out"] if last["out"] is not None else bn_act(z, st, last["x"])
    n = query("cova_colreduce_num_chunks", R, C)
    part = _empty((n, 2, C), z)
    call("cova_bn_bwd_reduce", dout, C, out, C, z, C, st.mean, st.invstd, R, C, part)
    g = torch.empty_like(dout)
    zero = torch.zeros((2, C), device=z.device)
    # coef = 0 and scale = 1 turn the apply kernel into the plain ReLU mask: dres = dout * (out > 0)
    call("cova_bn_bwd_apply", dout, C, out, C, z, C, st.mean, st.invstd, torch.ones_like(st.scale), zero,
         torch.empty_like(dout), C, g, C, R, C)
    return g, (part, n)


@on_device_of(1)
def convstack_bwd(sv, dfeat, gout=None, head_part=None, params=None, want_dimg=False):
    """dfeat NHWC [B,Hf,Wf,C] -> {state_dict key: grad} for the convs and BatchNorms of the stack; with ``want_dimg`` also
    the gradient w.r.t. the images (NCHW) under the key "__images__" (needs ``params``: conv1's weight).

    The data-gradient convs carry the ReLU mask and the BatchNorm-backward reduction of the layer
    in front of them in their epilogue (and its finalize in their tail), so only the last block's bn2
    (whose incoming gradient is RoIPool's scatter) needs a stand-alone reduction / finalize.
    ``params`` is needed by the resnet50 extension (its 1x1 convs read the weights directly) and by ``want_dimg``
    (conv1's weight)."""
    if want_dimg and (params is None or "convnet.0.weight" not in params):
        raise ValueError("convstack_bwd(want_dimg=True) needs params with 'convnet.0.weight' (the transposed conv1)")
    if sv["kind"] == "bottleneck" and params is None:
        raise ValueError("convstack_bwd of the resnet50 stack needs params (its 1x1 convolutions read the weights)")
    B, H, W, H1, W1, H2, W2 = sv["dims"]
    R = B * H2 * W2
    grads = {}
    if sv["kind"] == "bottleneck":
        if head_part is None:          # piecewise API (_get_visual_features): mask + sums stand-alone
            dfeat, head_part = masked_grad_and_sums(dfeat, sv["last"], R)
        dA = _layer1_bottleneck_bwd(sv, dfeat, params, gout, grads, head_part)
    else:
        dA = _layer1_bwd_fused(sv, dfeat, gout, grads, head_part)
    # maxpool + relu + bn1, then conv1's weight gradient
    bn1 = sv["bn1"]
    ws1 = _empty((query("cova_conv1_wgrad_workspace_floats", B, H, W),), dfeat)
    dw1 = _gbuf(gout, "convnet.0.weight", (64, 3, 7, 7), dfeat)
    pool_abc = sv.get("pool_abc")
    if pool_abc is None and sv.get("pool_part") is not None:        # (resnet50 stack: its last launch left the partials)
        dg = _gbuf(gout, "convnet.1.weight", (C64,), dfeat)
        db = _gbuf(gout, "convnet.1.bias", (C64,), dfeat)
        pool_abc = (dg, db, bn_finalize_bwd(sv["pool_part"], sv["pool_npart"], C64, B * H1 * W1, dg, db, "pages",
                                            abc_from=bn1, frozen=bn1.frozen))
    if pool_abc is not None:
        # dA is already ReLU-masked (epilogue of the last data-gradient conv): the pooling/BN backward
        # apply is folded into conv1's weight-gradient kernel, dy1 is never written
        dg, db, abc = pool_abc
        call("cova_conv1_wgrad_poolbwd", sv["images"], sv["y1"], dA, sv["idx"], abc, dw1, ws1, B, H, W)
        if want_dimg:                   # images.requires_grad: dy1 written out once, then the transposed convolution
            dy1 = torch.empty_like(sv["y1"])
            call("cova_pool_bwd_dy1", dA, sv["idx"], sv["y1"], abc, dy1, B, H1, W1)
    else:
        npart = query("cova_bn_relu_maxpool_bwd_num_partials", B, H1, W1)
        part = _empty((npart, 2, C64), dfeat)
        call("cova_bn_relu_maxpool_bwd_reduce", dA, sv["idx"], sv["y1"], bn1.scale, bn1.shift, bn1.mean,
             bn1.invstd, part, B, H1, W1)
        dg = _gbuf(gout, "convnet.1.weight", (C64,), dfeat)
        db = _gbuf(gout, "convnet.1.bias", (C64,), dfeat)
        coef = bn_finalize_bwd(part, npart, C64, B * H1 * W1, dg, db, "pages", frozen=bn1.frozen)
        dy1 = torch.empty_like(sv["y1"])
        call("cova_bn_relu_maxpool_bwd_apply", dA, sv["idx"], sv["y1"], bn1.scale, bn1.shift, bn1.mean,
             bn1.invstd, coef, dy1, B, H1, W1)
        call("cova_conv1_wgrad", sv["images"], dy1, dw1, ws1, B, H, W)
    grads["convnet.1.weight"], grads["convnet.1.bias"] = dg, db
    grads["convnet.0.weight"] = dw1
    if want_dimg:
        dimg = torch.empty_like(sv["images"])
        call("cova_conv1_dgrad", dy1, params["convnet.0.weight"], dimg, B, H, W)
        grads["__images__"] = dimg
    return grads


# ------------------------------------------------------------------------------- RoIPool
def roipool_fwd(feat, bboxes, roi_size, spatial_scale, out, ld_out):
    """feat: NHWC tensor or a LazyFeature (then bn2 + residual + ReLU are formed inside the kernel)."""
    _check(bboxes)
    B, Hf, Wf, C = feat.shape
    N = bboxes.shape[0]
    PH, PW = roi_size
    argmax = _empty((N, C * PH * PW), bboxes, torch.int32)
    zmax = None
    if isinstance(feat, LazyFeature):
        zmax = _empty((N, C * PH * PW), bboxes)
        call("cova_roipool_fwd_bn", feat.z, feat.x, feat.scale, feat.shift, bboxes, N, B, C, Hf, Wf, PH, PW,
             float(spatial_scale), out, ld_out, argmax, zmax)
    else:
        call("cova_roipool_fwd", feat, bboxes, N, B, C, Hf, Wf, PH, PW, float(spatial_scale), out, ld_out,
             argmax)
    return dict(argmax=argmax, bboxes=bboxes, shape=(B, Hf, Wf, C), roi=(PH, PW), scale=float(spatial_scale),
                zmax=zmax, pooled=out, ld_pooled=ld_out)


def roialign_fwd(feat, bboxes, roi_size, spatial_scale, sampling_ratio, aligned, out, ld_out):
    """RoIAlign variant of the pooling stage (extension: north_star names RoIAlign, the reference uses RoIPool)."""
    _check(bboxes)
    B, Hf, Wf, C = feat.shape
    N = bboxes.shape[0]
    PH, PW = roi_size
    call("cova_roialign_fwd", feat, bboxes, N, B, C, Hf, Wf, PH, PW, float(spatial_scale), int(sampling_ratio),
         1 if aligned else 0, out, ld_out)
    return dict(kind="align", bboxes=bboxes, shape=(B, Hf, Wf, C), roi=(PH, PW), scale=float(spatial_scale),
                sampling_ratio=int(sampling_ratio), aligned=bool(aligned), zmax=None)


def roialign_bwd(sv, gout, ld_g):
    B, Hf, Wf, C = sv["shape"]
    PH, PW = sv["roi"]
    gfeat = _empty((B, Hf, Wf, C), gout)
    call("cova_roialign_bwd", gout, ld_g, sv["bboxes"], sv["bboxes"].shape[0], B, C, Hf, Wf, PH, PW, sv["scale"],
         sv["sampling_ratio"], 1 if sv["aligned"] else 0, gfeat, _empty((2 * B + 16,), gout, torch.int32))
    return gfeat


def _roipool_ws(sv, like):
    B, Hf, Wf, C = sv["shape"]
    PH, PW = sv["roi"]
    return _empty((query("cova_roipool_bwd_workspace_words", sv["bboxes"].shape[0], B, C, PH, PW),), like, torch.int32)


def roipool_bwd(sv, gout, ld_g):
    B, Hf, Wf, C = sv["shape"]
    PH, PW = sv["roi"]
    gfeat = _empty((B, Hf, Wf, C), gout)
    call("cova_roipool_bwd", gout, ld_g, sv["bboxes"], sv["argmax"], sv["bboxes"].shape[0], B, C, Hf,
         Wf, PH, PW, float(sv["scale"]), gfeat, _roipool_ws(sv, gout))
    return gfeat


def roipool_bwd_bn(sv, gout, ld_g, last, grad_out=None):
    """RoIPool backward for the un-materialised feature map relu(bn(z) + x): the ReLU mask is `pooled > 0`
    (the pooled value is the map's value at the arg-max) and the BatchNorm-backward sums are taken per
    pooled entry from the z values the forward kept -> (masked gradient map, (partials, count)), or -- 64-channel
    stack, BatchNorm tails on -- (map, {"pend": (dgamma, dbeta, abc)}) with that BatchNorm's finalize done by the
    entry pass's last block."""
    B, Hf, Wf, C = sv["shape"]
    PH, PW = sv["roi"]
    n = sv["bboxes"].shape[0]
    gfeat = _empty((B, Hf, Wf, C), gout)
    npart = query("cova_roipool_bwd_bn_num_partials", n)
    part = _empty((npart, 2, C), gout)
    bn = last["bn"]
    if tails_on(C) and not bn.frozen and last.get("prefix"):
        dg, db, abc, tail = bn_tail_bwd(bn, B * Hf * Wf, grad_out, last["prefix"], gout)
        call("cova_roipool_bwd_bn_tail", gout, ld_g, sv["pooled"], sv["ld_pooled"], sv["zmax"], sv["bboxes"],
             sv["argmax"], n, B, C, Hf, Wf, PH, PW, float(sv["scale"]), bn.mean, bn.invstd, gfeat, part,
             _roipool_ws(sv, gout), tail.ptr)
        return gfeat, {"pend": (dg, db, abc)}
    call("cova_roipool_bwd_bn", gout, ld_g, sv["pooled"], sv["ld_pooled"], sv["zmax"], sv["bboxes"], sv["argmax"],
         n, B, C, Hf, Wf, PH, PW, float(sv["scale"]), bn.mean, bn.invstd, gfeat, part, _roipool_ws(sv, gout))
    return gfeat, (part, npart)


# ------------------------------------------------------------------------------- BN over [N, C]
def bn1d_fused(training):
    """one launch per BatchNorm1d application (csrc/bn1d.hip): train mode without SyncBN"""
    return OPTIONS.bn_tail and training and STAT_SYNC is None


def bn1d_fwd(x, ldx, N, C, prefix, params, buffers, training, out, ldo, relu, drop=None):
    """BatchNorm1d (+ReLU) of x [N,C] into out.  ``drop`` = (p, seed, mask or None): also the Dropout of the result
    -> returns (BNState, dropped, mask)."""
    if bn1d_fused(training) and N > 0:
        st = bn_state(C, x, N, True)
        dropped = mask = None
        p, seed, given = 0.0, 0, False
        if drop is not None:
            p, seed, mask = drop
            given = mask is not None
            if given:
                _check(mask, torch.uint8)
            else:
                mask = _empty((N, C), x, torch.uint8)
            dropped = _empty((N, C), x)
        call("cova_bn1d_fwd", x, ldx, N, C, params[prefix + "weight"], params[prefix + "bias"],
             buffers[prefix + "running_mean"], buffers[prefix + "running_var"],
             buffers.get(prefix + "num_batches_tracked"), BN_MOMENTUM, BN_EPS, 1 if relu else 0, out, ldo, dropped, C,
             mask, float(p), int(seed), 1 if given else 0, st.scale, st.shift, st.mean, st.invstd)
        return (st, dropped, mask) if drop is not None else st
    part, n = colstats(x, ldx, N, C) if training else (None, 0)
    st = bn_params(prefix, params, buffers, C, x, training, part, n, N)
    call("cova_bn_act_fwd", x, ldx, st.scale, st.shift, None, 0, out, ldo, N, C, 1 if relu else 0)
    if drop is not None:
        dropped, mask = dropout_fwd(out, ldo, N, C, drop[0], drop[1], drop[2])
        return st, dropped, mask
    return st


# ------------------------------------------------------------------------------- positional encoder
def bbox_fwd(bboxes, params, buffers, training, out, ldo):
    """models.py:129-148: [x1,y1,w,h,w/h] -> Linear(5,Hd) -> BatchNorm1d -> ReLU, written to out."""
    N = bboxes.shape[0]
    Hd = params["bbox_feat_encoder.0.weight"].shape[0]
    raw, z = _empty((N, 5), bboxes), _empty((N, Hd), bboxes)
    call("cova_bbox_linear_fwd", bboxes, params["bbox_feat_encoder.0.weight"],
         params["bbox_feat_encoder.0.bias"], raw, z, N, Hd)
    st = bn1d_fwd(z, Hd, N, Hd, "bbox_feat_encoder.1.", params, buffers, training, out, ldo, True)
    return dict(raw=raw, z=z, st=st, N=N, Hd=Hd, out=out, ldo=ldo)


def bbox_bwd(sv, g, ldg, gout=None):
    N, Hd = sv["N"], sv["Hd"]
    dz = _empty((N, Hd), g)
    dg, db = bn_backward(g, ldg, sv["out"], sv["ldo"], sv["z"], Hd, sv["st"], N, dz, Hd, None, 0,
                         gout, "bbox_feat_encoder.1.")
    dW = _gbuf(gout, "bbox_feat_encoder.0.weight", (Hd, 5), g)
    dbias = _gbuf(gout, "bbox_feat_encoder.0.bias", (Hd,), g)
    call("cova_bbox_linear_bwd", dz, sv["raw"], dW, dbias, N, Hd)
    return {"bbox_feat_encoder.0.weight": dW, "bbox_feat_encoder.0.bias": dbias,
            "bbox_feat_encoder.1.weight": dg, "bbox_feat_encoder.1.bias": db}


# ------------------------------------------------------------------------------- GAT
def _adjacent(a, b):
    """b starts right where a ends AND both are views of ONE storage (the trainer's flat bucket): [a; b] is one row-major
    matrix.  Two separately allocated tensors that the allocator happened to place back to back do not count: the
    one-GEMM and the two-GEMM forms associate their sums differently (an ulp), and which one a model instance takes
    must not depend on where its parameters landed in memory (found with two instances of the drop-in module built from
    one state_dict: bit-different gradients, tests/test_model_gpu.py::test_gradient_with_respect_to_the_images)."""
    return (a.is_contiguous() and b.is_contiguous() and a.shape[1:] == b.shape[1:] and
            b.data_ptr() == a.data_ptr() + a.numel() * a.element_size() and
            a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr())


def gat_fwd(h, ldh, N, F, ctx, params, hprime, ldo, prefix="gat."):
    """models.py:171-212.  h rows at h + n*ldh (F values); hprime rows at hprime + n*ldo (D)."""
    _check(ctx, torch.int64)
    Wi, Wj = params[prefix + "W_i.weight"], params[prefix + "W_j.weight"]
    aw, ab = params[prefix + "attention_layer.weight"], params[prefix + "attention_layer.bias"]
    D, K = Wi.shape[0], ctx.shape[1]
    Wh = _empty((N, 2 * D), h)
    if _adjacent(Wi, Wj):        # one [2D, F] matrix (flat parameter bucket): both projections in one GEMM
        call("cova_sgemm", 0, 1, N, 2 * D, F, h, ldh, Wi, F, Wh, 2 * D, None, 0)
    else:
        call("cova_sgemm", 0, 1, N, D, F, h, ldh, Wi, F, Wh, 2 * D, None, 0)
        call("cova_sgemm", 0, 1, N, D, F, h, ldh, Wj, F, Wh[:, D:], 2 * D, None, 0)
    s, t, attn = _empty((N,), h), _empty((N,), h), _empty((N, K), h)
    call("cova_gat_fwd", Wh, 2 * D, aw, ab, ctx, N, K, D, LEAKY_SLOPE, s, t, attn, hprime, ldo)
    return dict(h=h, ldh=ldh, N=N, F=F, D=D, K=K, ctx=ctx, Wh=Wh, s=s, t=t, attn=attn, prefix=prefix)


# Backward of the neighbour gather: a deterministic gather through the transposed index (no float atomics).
_CSR_WS = {}


def gat_transpose(ctx):
    """Transposed neighbour index of one batch (shared by every head / layer / backward call of the step).  The
    workspace is kept per (device, stream, N, K): its counters are zeroed once and left zero by the kernels.
    The returned tensor IS that cached workspace: it is valid until the next gat_transpose call with the same
    (device, stream, N, K) -- use it within the step that built it (clone it to keep it longer)."""
    N, K = ctx.shape
    key = (ctx.device, torch.cuda.current_stream(ctx.device).cuda_stream, N, K)
    csr = _CSR_WS.get(key)
    if csr is None:
        if len(_CSR_WS) >= 8:
            _CSR_WS.clear()
        csr = _CSR_WS[key] = torch.zeros((query("cova_gat_transpose_ints", N, K),), dtype=torch.int32, device=ctx.device)
    try:
        call("cova_gat_transpose_reuse", ctx, N, K, csr)
    except BaseException:               # (KeyboardInterrupt between the three launches included)
        _CSR_WS.pop(key, None)          # its counters may be half-updated: never reuse it
        raise
    return csr


def gat_bwd(sv, g, ldg, params, dh, lddh, accumulate_dh, gout=None, csr=None):
    """g = dL/dh' (rows at g + n*ldg).  Writes / accumulates dL/dh into dh; returns param grads."""
    N, F, D, K, prefix = sv["N"], sv["F"], sv["D"], sv["K"], sv["prefix"]
    if csr is None:
        csr = gat_transpose(sv["ctx"])
    du = _empty((N, K), g)
    Wi, Wj = params[prefix + "W_i.weight"], params[prefix + "W_j.weight"]
    aw = params[prefix + "attention_layer.weight"]
    dWh = _empty((N, 2 * D), g)
    ds, dt = _empty((N,), g), _empty((N,), g)
    daw = _gbuf(gout, prefix + "attention_layer.weight", (1, 2 * D), g)
    dab = _gbuf(gout, prefix + "attention_layer.bias", (1,), g)
    call("cova_gat_bwd", g, ldg, sv["Wh"], 2 * D, sv["s"], sv["t"], sv["attn"], sv["ctx"], aw, N, K, D,
         LEAKY_SLOPE, dWh, 2 * D, ds, dt, daw, dab, csr, du)
    dWi = _gbuf(gout, prefix + "W_i.weight", (D, F), g)
    dWj = _gbuf(gout, prefix + "W_j.weight", (D, F), g)
    if _adjacent(dWi, dWj):
        call("cova_sgemm", 1, 0, 2 * D, F, N, dWh, 2 * D, sv["h"], sv["ldh"], dWi, F, None, 0)
    else:
        call("cova_sgemm", 1, 0, D, F, N, dWh, 2 * D, sv["h"], sv["ldh"], dWi, F, None, 0)
        call("cova_sgemm", 1, 0, D, F, N, dWh[:, D:], 2 * D, sv["h"], sv["ldh"], dWj, F, None, 0)
    if _adjacent(Wi, Wj):
        call("cova_sgemm", 0, 0, N, F, 2 * D, dWh, 2 * D, Wi, F, dh, lddh, None, 1 if accumulate_dh else 0)
    else:
        call("cova_sgemm", 0, 0, N, F, D, dWh, 2 * D, Wi, F, dh, lddh, None, 1 if accumulate_dh else 0)
        call("cova_sgemm", 0, 0, N, F, D, dWh[:, D:], 2 * D, Wj, F, dh, lddh, None, 1)
    return {prefix + "W_i.weight": dWi, prefix + "W_j.weight": dWj,
            prefix + "attention_layer.weight": daw, prefix + "attention_layer.bias": dab}


def gat_stack_fwd(comb, T, N, F, D, ctx, params, n_heads=1, n_gat_layers=1):
    """The reference's single GraphAttentionLayer (models.py:114) or the multi-head / stacked extension:
    every layer concatenates its heads (hidden_dim/n_heads channels each, written side by side), layers
    are chained; the last one writes the context columns comb[:, F:]."""
    from .weights import gat_prefixes
    prefixes = gat_prefixes(n_heads, n_gat_layers)
    dh = D // n_heads
    layers, h, ldh, fin = [], comb, T, F
    for l, heads in enumerate(prefixes):
        last = l == len(prefixes) - 1
        out, ldo = (comb[:, F:], T) if last else (_empty((N, D), comb), D)
        svs = [gat_fwd(h, ldh, N, fin, ctx, params, out[:, i * dh:], ldo, prefix=p)
               for i, p in enumerate(heads)]
        layers.append(dict(heads=svs, out=out))
        h, ldh, fin = out, ldo, D
    return layers


def gat_stack_bwd(layers, dcomb, T, N, F, D, params, gout=None):
    """dcomb[:, F:] = dL/d(context); accumulates dL/d(own features) into dcomb[:, :F]."""
    grads = {}
    g, ldg = dcomb[:, F:], T
    csr = gat_transpose(layers[0]["heads"][0]["ctx"])
    for l in reversed(range(len(layers))):
        heads = layers[l]["heads"]
        dh = D // len(heads)
        if l == 0:
            dst, ldd, acc0 = dcomb, T, True
        else:
            dst, ldd, acc0 = _empty((N, D), dcomb), D, False
        for i, sv in enumerate(heads):
            grads.update(gat_bwd(sv, g[:, i * dh:], ldg, params, dst, ldd, acc0 or i > 0, gout, csr))
        g, ldg = dst, ldd
    return grads


# ------------------------------------------------------------------------------- decoder
def dropout_fwd(x, ld, N, C, p, seed, mask=None):
    out = _empty((N, C), x)
    given = mask is not None
    if not given:
        mask = _empty((N, C), x, torch.uint8)
    else:
        _check(mask, torch.uint8)
    call("cova_dropout_fwd", x, ld, out, C, mask, N, C, float(p), int(seed), 1 if given else 0)
    return out, mask


def decoder_fwd(x, N, T, params, buffers, training, p, seeds=(0, 0), masks=None):
    """models.py:83-90 on the concatenated features x [N,T]: Dropout, Linear, BN1d, ReLU, Dropout,
    Linear.  ``masks`` (two uint8 [N,T] keep-masks) override the generated ones (parity tests)."""
    NC = params["decoder.5.weight"].shape[0]
    drop = training and (p > 0 or masks is not None)
    sv = dict(N=N, T=T, NC=NC, drop=drop, p=p)
    xd, m1 = dropout_fwd(x, T, N, T, p, seeds[0], masks[0] if masks else None) if drop else (x, None)
    z = _empty((N, T), x)
    call("cova_sgemm", 0, 1, N, T, T, xd, T, params["decoder.1.weight"], T, z, T,
         params["decoder.1.bias"], 0)
    y = _empty((N, T), x)
    if drop:
        st, yd, m2 = bn1d_fwd(z, T, N, T, "decoder.2.", params, buffers, training, y, T, True,
                              drop=(p, seeds[1], masks[1] if masks else None))
    else:
        st, yd, m2 = bn1d_fwd(z, T, N, T, "decoder.2.", params, buffers, training, y, T, True), y, None
    logits = _empty((N, NC), x)
    call("cova_linear_small_fwd", yd, T, params["decoder.5.weight"], params["decoder.5.bias"], logits,
         N, T, NC)
    sv.update(xd=xd, m1=m1, z=z, y=y, st=st, yd=yd, m2=m2)
    return logits, sv


def decoder_bwd(sv, dlogits, params, gout=None):
    """-> (dL/dx [N,T], param grads)."""
    N, T, NC, p = sv["N"], sv["T"], sv["NC"], sv["p"]
    _check(dlogits)
    dyd = _empty((N, T), dlogits)
    dW2 = _gbuf(gout, "decoder.5.weight", (NC, T), dlogits)
    db2 = _gbuf(gout, "decoder.5.bias", (NC,), dlogits)
    call("cova_linear_small_bwd", dlogits, sv["yd"], T, params["decoder.5.weight"], dyd, T, dW2, db2,
         N, T, NC)
    dz = _empty((N, T), dlogits)
    db1 = _gbuf(gout, "decoder.1.bias", (T,), dlogits)
    dg, db = bn_backward(dyd, T, sv["y"], T, sv["z"], T, sv["st"], N, dz, T, None, 0, gout, "decoder.2.",
                         drop=(sv["m2"], p) if sv["drop"] else None, colsum=db1)
    dW1 = _gbuf(gout, "decoder.1.weight", (T, T), dlogits)
    call("cova_sgemm", 1, 0, T, T, N, dz, T, sv["xd"], T, dW1, T, None, 0)
    dx = dyd        # (dead by now: reuse)
    if sv["drop"]:  # the first Dropout's backward in the GEMM's epilogue (one launch less, same bits)
        call("cova_sgemm_dropout_bwd", 0, 0, N, T, T, dz, T, params["decoder.1.weight"], T, dx, T, sv["m1"], float(p))
    else:
        call("cova_sgemm", 0, 0, N, T, T, dz, T, params["decoder.1.weight"], T, dx, T, None, 0)
    grads = {"decoder.1.weight": dW1, "decoder.1.bias": db1, "decoder.2.weight": dg,
             "decoder.2.bias": db, "decoder.5.weight": dW2, "decoder.5.bias": db2}
    return dx, grads


# ------------------------------------------------------------------------------- whole model
@on_device_of(3)
def model_fwd(cfg, params, buffers, images, bboxes, additional_feats, context_indices, training,
              seeds=(0, 0), masks=None, save=True):
    """CoVA.forward (models.py:94-122) -> (logits [N,n_classes], saved-for-backward or None)."""
    N = bboxes.shape[0]
    PH, PW = cfg["roi_output_size"]
    n_vis = (C256 if is_bottleneck(params) else C64) * PH * PW
    Hd, A = cfg["bbox_hidden_dim"], cfg["n_additional_feat"]
    F = n_vis + Hd + A
    D = cfg["hidden_dim"] if cfg["use_context"] else 0
    T = F + D
    align = cfg.get("roi_op", "pool") == "align"
    feat, sv_conv = convstack_fwd(images, params, buffers, training, save, lazy_out=not align)
    comb = _empty((N, T), images)
    scale = cfg.get("spatial_scale") or feat.shape[1] / images.shape[2]   # models.py:56
    sv = dict(cfg=cfg, N=N, F=F, D=D, T=T, n_vis=n_vis, Hd=Hd, A=A, conv=sv_conv, comb=comb)
    if align:
        sv["roi"] = roialign_fwd(feat, bboxes, (PH, PW), scale, cfg.get("sampling_ratio", 2),
                                 cfg.get("roi_aligned", False), comb, T)
    else:
        sv["roi"] = roipool_fwd(feat, bboxes, (PH, PW), scale, comb, T)
    if Hd > 0:
        sv["bbox"] = bbox_fwd(bboxes, params, buffers, training, comb[:, n_vis:], T)
    if A > 0:
        _check(additional_feats)
        sv["addl_in"] = additional_feats
        sv["addl"] = bn1d_fwd(additional_feats, A, N, A, "bn_additional_feat.", params, buffers,
                              training, comb[:, n_vis + Hd:], T, False)
    if D > 0:
        sv["gat"] = gat_stack_fwd(comb, T, N, F, D, context_indices, params, cfg.get("n_heads", 1),
                                  cfg.get("n_gat_layers", 1))
    logits, sv["dec"] = decoder_fwd(comb, N, T, params, buffers, training, cfg["drop_prob"], seeds,
                                    masks)
    return logits, (sv if save else None)


@on_device_of(1)
def model_bwd(sv, dlogits, params, gout=None, after_head=None, want_dimg=False):
    """-> {state_dict key: gradient} for every trainable parameter (+ "__images__" with ``want_dimg``).  ``gout`` (optional)
    maps keys to pre-allocated destinations, e.g. views into one flat all-reduce bucket.
    ``after_head`` (optional callable) runs once every gradient outside the conv stack is final
    (the trainer starts their all-reduce there, under the conv-stack backward)."""
    N, F, D, T, n_vis, Hd, A = (sv[k] for k in ("N", "F", "D", "T", "n_vis", "Hd", "A"))
    dcomb, grads = decoder_bwd(sv["dec"], dlogits, params, gout)
    if D > 0:
        grads.update(gat_stack_bwd(sv["gat"], dcomb, T, N, F, D, params, gout))
    if A > 0:
        st = sv["addl"]
        dz = _empty((N, A), dcomb)
        dg, db = bn_backward(dcomb[:, n_vis + Hd:], T, None, 0, sv["addl_in"], A, st, N, dz, A, None, 0,
                             gout, "bn_additional_feat.")
        grads["bn_additional_feat.weight"], grads["bn_additional_feat.bias"] = dg, db
    if Hd > 0:
        grads.update(bbox_bwd(sv["bbox"], dcomb[:, n_vis:], T, gout))
    if after_head is not None:
        after_head()
    conv = sv["conv"]
    if N > 0 and sv["roi"]["zmax"] is not None:
        dfeat, head_part = roipool_bwd_bn(sv["roi"], dcomb, T, conv["last"], gout)
        grads.update(convstack_bwd(conv, dfeat, gout, head_part, params, want_dimg))
    else:
        dfeat = roialign_bwd(sv["roi"], dcomb, T) if sv["roi"].get("kind") == "align" else roipool_bwd(sv["roi"], dcomb, T)
        grads.update(convstack_bwd(conv, dfeat, gout, None, params, want_dimg))
    return grads


def ce_sum(logits, labels, want_grad=True, gscale=1.0):
    """CrossEntropyLoss(reduction='sum') + argmax (main.py:139, train.py:53,56)."""
    N, NC = logits.shape
    loss = _empty((1,), logits)
    dl = _empty((N, NC), logits) if want_grad else None
    pred = _empty((N,), logits, torch.int64)
    call("cova_ce_sum", logits, labels, N, NC, float(gscale), loss, dl, pred)
    return loss, dl, pred
