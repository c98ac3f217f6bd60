"""Shared helpers for the parity tests (fixture loading, seeded regeneration of inputs)."""
import os

import numpy as np
import torch

import cova_amd  # noqa: F401
from cova_web_object_detection_amd import synthetic, weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_CASES = ["cova_h64_n11", "cova_h64_n90", "cova_h128_ragged", "cova_h64_addfeat"]
SAMPLE_STRIDE = 97


def load_case(name):
    """Golden fixture -> (fixture dict, cfg, state_dict, batch) with inputs regenerated from seeds."""
    fx = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    boxes = [int(b) for b in fx["meta/boxes"]]
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True,
               hidden_dim=int(fx["meta/hidden_dim"]), bbox_hidden_dim=int(fx["meta/bbox_hidden_dim"]),
               n_additional_feat=int(fx["meta/n_additional_feat"]), drop_prob=0.0)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(int(fx["meta/seed"]), logit_gain=float(fx["meta/logit_gain"]), **wcfg)
    batch = synthetic.make_batch(int(fx["meta/n_pages"]), img_h=int(fx["meta/img_h"]),
                                 boxes_per_page=boxes, context_size=int(fx["meta/context_size"]),
                                 n_additional_feat=cfg["n_additional_feat"], seed=int(fx["meta/seed"]))
    return fx, cfg, sd, batch


def check_grads(fx, grads, rtol, floor_frac=0.01):
    """Compare a {key: tensor} gradient dict with a fixture's grad/gradsample/gradnorm entries.

    Several parameters have analytically ZERO gradient (a bias in front of a train-mode
    BatchNorm, anything softmax/BN shift-invariant), so the tolerance for tensor k is
    ``rtol * max(max|ref_k|, floor_frac * max_k max|ref_k|)`` rather than purely relative.
    """
    keys = [k[len("gradnorm/"):] for k in fx if k.startswith("gradnorm/")]
    refs = {}
    for k in keys:
        refs[k] = fx["grad/" + k] if "grad/" + k in fx else fx["gradsample/" + k]
    gscale = max(float(np.abs(r).max()) for r in refs.values())
    worst = {}
    for k in keys:
        g = grads[k].detach().cpu().numpy().astype(np.float32)
        ref = refs[k]
        got = g if "grad/" + k in fx else g.reshape(-1)[::SAMPLE_STRIDE]
        scale = max(float(np.abs(ref).max()), floor_frac * gscale)
        err = float(np.abs(got.reshape(ref.shape) - ref).max()) / scale
        worst[k] = err
        assert err <= rtol, "grad %s: max err/scale %.3e > %.1e" % (k, err, rtol)
        nrm = float(np.linalg.norm(g.astype(np.float64)))
        ref_norm = float(fx["gradnorm/" + k])
        assert abs(nrm - ref_norm) <= rtol * max(ref_norm, floor_frac * gscale * np.sqrt(g.size)), \
            (k, nrm, ref_norm)
    return worst


def unforced_fraction_above(fx, grads, rtol=2e-4, floor_frac=0.01):
    """Fraction of the fixture's gradient ENTRIES (full tensors and strided samples, as stored) that differ from the
    implementation's by more than ``rtol`` of the tensor's scale, with NO decision forced: the direct, one-hop
    comparison with what the reference's autograd produced.  Returns (fraction, entries compared, worst tensor,
    per-tensor counts); the caller prints and bounds it."""
    keys = [k[len("gradnorm/"):] for k in fx if k.startswith("gradnorm/")]
    refs = {k: (fx["grad/" + k] if "grad/" + k in fx else fx["gradsample/" + k]) for k in keys}
    gscale = max(float(np.abs(r).max()) for r in refs.values())
    bad = total = 0
    per = {}
    for k, ref in refs.items():
        g = grads[k].detach().cpu().numpy().astype(np.float32)
        got = g if "grad/" + k in fx else g.reshape(-1)[::SAMPLE_STRIDE]
        scale = max(float(np.abs(ref).max()), floor_frac * gscale)
        d = np.abs(got.reshape(ref.shape) - ref) / scale
        n = int((d > rtol).sum())
        per[k] = (n, int(d.size), float(d.max()))
        bad += n
        total += int(d.size)
    worst = max(per, key=lambda k: per[k][2])
    return bad / max(total, 1), total, worst, per


def margins_ok(logits, tol):
    """Rows whose top-2 logit gap exceeds `tol` (integer predictions are asserted on these)."""
    top2 = torch.topk(logits, 2, dim=1).values
    return (top2[:, 0] - top2[:, 1]) > tol


def routing_from_saved(sv):
    """RoIPool argmax + max-pool indices chosen by the HIP forward, in the oracle's index format.

    fp32 implementations that sum in different orders disagree on near-tied maxima (a handful
    of the ~1e5 routing decisions per batch); gradient parity is asserted for identical routing
    and the disagreements themselves are asserted to be near-ties (assert_routing_near_ties)."""
    from oracle import cova_oracle as O
    from cova_web_object_detection_amd import engine as E
    B, H, W, H1, W1, H2, W2 = sv["conv"]["dims"]
    r = {"roi_argmax": sv["roi"]["argmax"].cpu(),
         "pool_idx": O.pool_window_pos_to_flat(sv["conv"]["idx"].cpu(), H1, W1)}
    # ReLU gates taken by the HIP forward (NHWC -> the oracle's NCHW)
    nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu()
    conv = sv["conv"]
    bn1 = conv["bn1"]
    r["gate_bn1"] = nchw(conv["y1"] * bn1.scale + bn1.shift > 0)
    for i, blk in enumerate(conv["blocks"]):
        if conv["kind"] == "bottleneck":
            r["gate_a1_%d" % i] = nchw(E.bn_act(blk["z1"], blk["bn1"]) > 0)
            r["gate_a2_%d" % i] = nchw(E.bn_act(blk["z2"], blk["bn2"]) > 0)
            out = blk["out"] if blk["out"] is not None else E.bn_act(blk["z3"], blk["bn3"], blk["x"])
            r["gate_out_%d" % i] = nchw(out > 0)
        else:
            r["gate_a1_%d" % i] = nchw(E.block_a1(blk) > 0)
            r["gate_out_%d" % i] = nchw(E.block_out(blk) > 0)
    n_vis, hd = sv["n_vis"], sv["Hd"]
    if hd > 0:
        r["gate_bbox"] = (sv["comb"][:, n_vis:n_vis + hd] > 0).cpu()
    r["gate_dec"] = (sv["dec"]["y"] > 0).cpu()
    # LeakyReLU slope decisions of every GAT head: the sign of u = s_i + t_j (pad slots: t = 0; they are masked anyway)
    for layer in sv.get("gat") or []:
        for head in layer["heads"]:
            ctx = head["ctx"]
            t = torch.cat((head["t"], head["t"].new_zeros(1)))
            u = head["s"].view(-1, 1) + t[ctx.clamp(min=-1)]
            r["gate_" + head["prefix"] + "leaky"] = (u > 0).cpu()
    return r


def assert_routing_near_ties(routing, inter, bboxes, roi_size, scale, rel_tol=1e-4, max_frac=2e-3):
    """Every RoIPool argmax that differs from the oracle's must point at a value within
    rel_tol*max|feat| of the oracle's maximum, and such flips must be rare."""
    from oracle import cova_oracle as O
    feat = inter["feat"].detach()
    out, arg = O.roi_pool_argmax(feat, bboxes, roi_size, scale)
    n = bboxes.shape[0]
    arg = arg.reshape(n, -1)
    got = routing["roi_argmax"].reshape(n, -1)
    diff = (arg != got)
    nflip = int(diff.sum())
    assert nflip <= max(2, max_frac * arg.numel()), "too many RoIPool routing flips: %d" % nflip
    if nflip:
        C = feat.shape[1]
        bins = roi_size[0] * roi_size[1]
        fmax = float(feat.abs().max())
        rows, cols = diff.nonzero(as_tuple=True)
        for r, c in zip(rows.tolist(), cols.tolist()):
            b, ch = int(bboxes[r, 0]), c // bins
            plane = feat[b, ch].reshape(-1)
            assert got[r, c] >= 0 and arg[r, c] >= 0
            assert abs(float(plane[got[r, c]]) - float(plane[arg[r, c]])) <= rel_tol * fmax
    return nflip


def compare_grads(grads, grads_ref, rtol, floor_frac=0.01, outlier_frac=2e-3, outlier_rtol=5e-2):
    """Element-wise gradient parity that is robust to discrete gate flips.

    Two fp32 implementations that sum in different orders legitimately disagree on a handful of
    discrete decisions per batch -- a ReLU gate whose pre-activation is within ~1e-6 of zero, a
    near-tied max -- and each such flip moves a few gradient entries by O(1e-3..1e-2) of the
    tensor's scale.  So: at most ``outlier_frac`` of a tensor's entries may exceed
    ``rtol*scale`` and none may exceed ``outlier_rtol*scale`` (a real kernel bug violates both
    massively).  scale = max(max|ref|, floor_frac * largest gradient) because several
    parameters have analytically zero gradient."""
    gscale = max(float(g.abs().max()) for g in grads_ref.values())
    worst = ("", 0.0)
    for k, g in grads_ref.items():
        scale = max(float(g.abs().max()), floor_frac * gscale)
        diff = (grads[k].detach().cpu().view_as(g) - g).abs() / scale
        err = float(diff.max())
        bad = int((diff > rtol).sum())
        if err > worst[1]:
            worst = (k, err)
        assert bad <= (max(2, outlier_frac * diff.numel()) if outlier_frac > 0 else 0), \
            "grad %s: %d of %d entries off by > %.1e" % (k, bad, diff.numel(), rtol)
        assert err < outlier_rtol, "grad %s: max err/scale %.3e >= %.1e" % (k, err, outlier_rtol)
    return worst


def assert_gate_flips_near_zero(routing, tap, rel_tol=2e-5, max_frac=1e-4):
    """Discrete decisions of the HIP forward that differ from the UNFORCED oracle's: every flipped ReLU gate
    must sit on a pre-activation within rel_tol * max|pre-activation| of zero, every max-pool arg-max that
    differs must point at a value within the same distance of the oracle's maximum, and both must be rare
    (<= max_frac of the decisions of that layer).  Returns {key: number of flips}."""
    flips = {}
    for key, pre in tap.items():
        if key == "pool_in" or key not in routing:
            continue
        gate = routing[key].to(torch.bool).reshape(pre.shape)
        diff = gate != (pre > 0)
        n = int(diff.sum())
        flips[key] = n
        assert n <= max(2, max_frac * diff.numel()), "%s: %d of %d ReLU gates differ" % (key, n, diff.numel())
        if n:
            worst = float(pre[diff].abs().max())
            assert worst <= rel_tol * float(pre.abs().max()), "%s: a flipped gate sits at %.3e" % (key, worst)
    if "pool_in" in tap and "pool_idx" in routing:
        x = tap["pool_in"]
        B, C, H, W = x.shape
        ref = torch.nn.functional.max_pool2d(x, 3, 2, 1, return_indices=True)[1]
        got = routing["pool_idx"].reshape(ref.shape)
        diff = got != ref
        n = int(diff.sum())
        flips["pool_idx"] = n
        assert n <= max(2, max_frac * diff.numel()), "max-pool: %d of %d arg-maxes differ" % (n, diff.numel())
        if n:
            flat = x.reshape(B, C, H * W)
            a = torch.gather(flat, 2, got.reshape(B, C, -1))[diff.reshape(B, C, -1)]
            b = torch.gather(flat, 2, ref.reshape(B, C, -1))[diff.reshape(B, C, -1)]
            assert float((a - b).abs().max()) <= rel_tol * float(x.abs().max())
    return flips
