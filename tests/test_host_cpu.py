"""CPU tests of the host logic: C-ABI library loads and exports every declared symbol, the
drop-in module carries the reference's state_dict contract, and nothing falls back to CPU."""
import ctypes
import os

import numpy as np
import pytest
import torch

from cova_web_object_detection_amd import _lib, synthetic, weights
from cova_web_object_detection_amd.models import CoVA, GraphAttentionLayer


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 35
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), name
    # pure host queries work without a GPU
    assert _lib.query("cova_conv_out_size", 1280, 7, 2, 3) == 640
    assert _lib.query("cova_conv_out_size", 640, 3, 2, 1) == 320
    assert _lib.query("cova_conv3x3_wino4_num_tiles", 16, 320, 320) == 16 * 40 * 10


def test_header_has_no_torch_types():
    src = open(_lib.HEADER).read()
    assert "torch" not in src.replace("torch.cuda.current_stream", "").replace("torch /", "").lower() \
        or "at::" not in src
    assert "at::Tensor" not in src and "#include <torch" not in src


def test_module_state_dict_contract():
    m = CoVA((3, 3), 1280, 4, True, 384, 32, 0, 0.2, None)
    spec = weights.state_dict_spec()
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in spec]
    for k, shape in spec:
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in m.parameters()) == 1616485
    assert abs(m.roi_pool.spatial_scale - 0.25) < 1e-12
    m.load_state_dict(weights.seeded_state_dict(3))
    m2 = CoVA((3, 3), 1280, 4, True, 384, 32, 7, 0.2, None)      # CoVA++ branch
    assert "bn_additional_feat.weight" in m2.state_dict() and m2.n_feat == 615


def test_no_cpu_fallback():
    m = CoVA((3, 3), 64, 4)
    b = synthetic.make_batch(1, img_h=64, boxes_per_page=5, seed=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(b["images"], b["bboxes"], b["additional_feats"], b["context_indices"])
    layer = GraphAttentionLayer(8, 4)
    with pytest.raises(RuntimeError):
        layer(torch.zeros(3, 8), torch.zeros(3, 2, dtype=torch.long))


def test_synthetic_batch_contract():
    b = synthetic.make_batch(3, img_h=32, boxes_per_page=[11, 40, 25], context_size=12, seed=5)
    assert b["images"].shape == (3, 3, 32, 32) and b["bboxes"].shape == (76, 5)
    assert b["context_indices"].shape == (76, 24) and b["context_indices"].dtype == torch.int64
    ctx, idx = b["context_indices"].numpy(), b["bboxes"][:, 0].numpy()
    for i in range(76):
        nb = ctx[i][ctx[i] >= 0]
        assert (idx[nb] == idx[i]).all()                 # neighbours never cross pages
        assert i not in nb
    for p in range(3):
        lab = b["labels"][idx == p].numpy()
        assert sorted(lab[lab > 0].tolist()) == [1, 2, 3]
    b2 = synthetic.make_batch(3, img_h=32, boxes_per_page=[11, 40, 25], context_size=12, seed=5)
    assert all(torch.equal(b[k], b2[k]) for k in b if torch.is_tensor(b[k]))


def test_extension_modules_keep_the_reference_surface():
    """backbone / n_heads / n_gat_layers are defaulted extras: the 9 positional reference arguments are
    unchanged, the extension state_dict follows weights.state_dict_spec, the GAT container is callable
    like GraphAttentionLayer."""
    import inspect
    import warnings
    names = list(inspect.signature(CoVA.__init__).parameters)[1:10]
    assert names == ["roi_output_size", "img_H", "n_classes", "use_context", "hidden_dim", "bbox_hidden_dim",
                     "n_additional_feat", "drop_prob", "class_names"]
    for kw, nparam in ((dict(backbone="resnet50", n_heads=2), 9437862),
                       (dict(backbone="resnet50", n_heads=2, n_gat_layers=2), 9733544)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = CoVA((3, 3), 1280, 4, True, 384, 32, 0, 0.2, None, **kw)
        spec = weights.state_dict_spec(**kw)
        assert list(m.state_dict().keys()) == [k for k, _ in spec]
        assert sum(p.numel() for p in m.parameters()) == nparam
        assert m.n_visual_feat == 2304 and m.n_total_feat == 2720
        m.load_state_dict(weights.seeded_state_dict(1, **kw))
    CoVA._warned_random_init.clear()                 # the warning is issued once per process and architecture
    with pytest.warns(UserWarning, match="ImageNet"):
        CoVA((3, 3), 64, 4)
    with warnings.catch_warnings():
        warnings.simplefilter("error")               # ... and not again
        CoVA((3, 3), 64, 4)
    # torchvision-style backbone weights initialise conv1 / bn1 / layer1 (offline stand-in for pretrained=True)
    sd = weights.seeded_state_dict(4)
    tv = {}
    for k, v in sd.items():
        for dst, src in (("conv1.", "convnet.0."), ("bn1.", "convnet.1."), ("layer1.", "convnet.4.")):
            if k.startswith(src):
                tv[dst + k[len(src):]] = v
    tv["fc.weight"] = torch.zeros(3, 3)                       # entries outside the truncated stack are ignored
    m = CoVA((3, 3), 64, 4, backbone_state_dict=tv)
    assert torch.equal(m.state_dict()["convnet.4.1.conv2.weight"], sd["convnet.4.1.conv2.weight"])


def test_shard_batch_and_evaluate_bookkeeping_long_pages():
    from cova_web_object_detection_amd.trainer import shard_batch
    b = synthetic.make_batch(4, img_h=32, img_w=16, boxes_per_page=[300, 5, 61, 300], context_size=24, seed=2)
    assert b["context_indices"].shape[1] == 48 and b["images"].shape == (4, 3, 32, 16)
    tot = 0
    for r in range(2):
        s = shard_batch(b, r, 2)
        n = s["bboxes"].shape[0]
        tot += n
        assert int(s["context_indices"].max()) < n and int(s["bboxes"][:, 0].max()) == 1
    assert tot == 666


def test_bench_flop_accounting_matches_survey_8d():
    """bench.py's algorithmic-work model reproduces SURVEY.md 8d's per-page figures (the numbers `step.*` and the
    roofline are derived from): 106.8 GF (R18, 1280^2), 151.3 GF (R50 stem), 484.7 GF (R50, 1280x4096, n=300, K=48)."""
    import bench
    for cfg, gf in ((2, 106.8), (3, 151.3), (5, 484.7)):
        fm = bench.flop_model(bench.WORKLOADS[cfg])
        assert abs(fm["total"] / 1e9 - gf) < 0.012 * gf, (cfg, fm["total"] / 1e9)
    fm = bench.flop_model(bench.WORKLOADS[2])
    assert fm["conv3_launch_per_page"] == 2 * 64 * 64 * 9 * 320 * 320
    assert fm["wino"] == 12 * fm["conv3_launch_per_page"]
    assert bench.CFG["backbone"] == "resnet18" and bench.WORKLOADS[4]["pages"] == 32


def test_trainer_checkpoint_round_trip_on_cpu_buffers():
    """HotPathTrainer's flat buckets are device agnostic: the save-best / reload-best cycle of train.py:84,94 and an
    optimizer-state round trip can be checked without a GPU (no device work is launched)."""
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=32, bbox_hidden_dim=8,
               n_additional_feat=0, drop_prob=0.2)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd_a, sd_b = weights.seeded_state_dict(1, **wcfg), weights.seeded_state_dict(2, **wcfg)
    tr = HotPathTrainer(cfg, sd_a, "cpu")
    ptr = tr.pbucket.flat.data_ptr()
    got = tr.state_dict()
    assert list(got.keys()) != [] and all(torch.equal(got[k], sd_a[k]) for k in sd_a)
    tr.load_state_dict(sd_b)
    assert tr.pbucket.flat.data_ptr() == ptr and all(torch.equal(tr.state_dict()[k], sd_b[k]) for k in sd_b)
    tr.exp_avg.fill_(0.5)
    tr.step_count = 17
    st = tr.optimizer_state_dict()
    tr2 = HotPathTrainer(cfg, sd_b, "cpu")
    tr2.load_optimizer_state_dict(st)
    assert tr2.step_count == 17 and torch.equal(tr2.exp_avg, tr.exp_avg)
    with pytest.raises(KeyError):
        tr.load_state_dict({"convnet.0.weight": sd_a["convnet.0.weight"]})


def test_flat_bucket_layout_of_the_extension_model():
    """One flat parameter / gradient bucket also for the resnet50 + multi-head spec: every parameter has a 16-byte
    aligned view, conv-stack tensors come first (the two-phase all-reduce splits there), buffers stay outside."""
    from cova_web_object_detection_amd.trainer import HotPathTrainer, is_param_key
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=64, bbox_hidden_dim=8,
               n_additional_feat=0, drop_prob=0.2, backbone="resnet50", n_heads=2, n_gat_layers=2)
    sd = weights.seeded_state_dict(1, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    tr = HotPathTrainer(cfg, sd, "cpu")
    pkeys = [k for k in sd if is_param_key(k)]
    assert list(tr.params) == pkeys and set(tr.buffers) == set(sd) - set(pkeys)
    for k, (o, m, shape) in tr.gbucket.offsets.items():
        assert o % 4 == 0 and m == sd[k].numel() and tuple(shape) == tuple(sd[k].shape)
        assert torch.equal(tr.params[k], sd[k])
    head = tr._head_offset()
    conv = [k for k in pkeys if k.startswith("convnet.")]
    assert head == sum((sd[k].numel() + 3) // 4 * 4 for k in conv)
    assert pkeys[len(conv)].startswith("bbox_feat_encoder.") and "gat.layers.1.heads.1.W_j.weight" in tr.params


def test_launch_device_resolution_rejects_cpu_and_mixed_arguments():
    """_lib.call launches on the device its tensor arguments live on (the reference picks cuda:<-d> without
    set_device, main.py:17); CPU tensors and mixed devices are refused before anything is launched."""
    with pytest.raises(_lib.CovaHipError, match="no CPU fallback"):
        _lib._device_of("cova_x", (torch.zeros(2), 3, None))
    assert _lib._device_of("cova_x", (3, None, 1.5)) is None

    class Fake:                                   # two "cuda" tensors on different devices, without a GPU
        def __init__(self, idx):
            self.is_cuda, self.device = True, torch.device("cuda", idx)
    orig = torch.Tensor
    try:
        _lib.torch.Tensor = Fake                  # isinstance check inside _device_of
        with pytest.raises(_lib.CovaHipError, match="different devices"):
            _lib._device_of("cova_x", (Fake(0), Fake(1)))
        assert _lib._device_of("cova_x", (Fake(1), Fake(1))) == torch.device("cuda", 1)
    finally:
        _lib.torch.Tensor = orig


def test_check_batch_host_contract():
    """engine.check_batch is pure host logic: the shape errors of the reference's forward, no device read."""
    from cova_web_object_detection_amd import engine
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32,
               n_additional_feat=0, drop_prob=0.2)
    img, bb, af = torch.zeros(2, 3, 8, 8), torch.zeros(6, 5), torch.zeros(6, 0)
    ctx = torch.zeros(6, 4, dtype=torch.long)
    engine.check_batch(cfg, img, bb, af, ctx, True)
    engine.check_batch(cfg, img, bb, af, ctx.int(), False)
    engine.check_batch(dict(cfg, use_context=False), img, bb, af, torch.empty(0, 0, dtype=torch.long), True)
    engine.check_batch(cfg, img, bb[:0], af[:0], ctx[:0], True)                 # empty batch is legal
    engine.check_batch(cfg, img, bb[:1], af[:1], ctx[:1], False)                # one box in eval mode too
    for bad in ((img[:, :2], bb, af, ctx), (img, bb[:, :4], af, ctx), (img, bb, torch.zeros(6, 2), ctx),
                (img, bb, af, ctx[:3]), (img, bb, af, ctx[0])):
        with pytest.raises(RuntimeError):
            engine.check_batch(cfg, *bad, True)
    with pytest.raises(IndexError):
        engine.check_batch(cfg, img, bb, af, ctx.float(), True)
    with pytest.raises(ValueError, match=r"torch.Size\(\[1, 32\]\)"):
        engine.check_batch(cfg, img, bb[:1], af[:1], ctx[:1], True)
    with pytest.raises(ValueError, match=r"torch.Size\(\[1, 960\]\)"):     # no positional encoder: decoder BN
        engine.check_batch(dict(cfg, bbox_hidden_dim=0), img, bb[:1], af[:1], ctx[:1], True)
    engine.check_batch(cfg, img, bb, af, torch.zeros(6, 100, dtype=torch.long), True)      # -cs 50: beyond one wavefront
    with pytest.raises(ValueError):
        engine.check_batch(cfg, img, bb, af, torch.zeros(6, engine.GAT_MAX_K + 1, dtype=torch.long), True)
