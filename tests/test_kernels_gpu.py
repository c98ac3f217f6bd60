"""GPU parity tests, kernel level: every C-ABI entry point against torch-CPU fp32 / the oracle.

Tolerances: fp32 everywhere; conv / GEMM reductions up to K = 992 reorder the summation, so
values are compared with atol = rtol = 1e-4 relative to the tensor's magnitude (2e-4 for
gradients that sum over >1e4 pixels).  Integer outputs (RoIPool argmax, pool indices, argmax
predictions, dropout masks) are compared exactly.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cova_web_object_detection_amd import _lib, engine, synthetic  # noqa: E402
from oracle import cova_oracle as O  # noqa: E402

call, query = _lib.call, _lib.query
DEV = "cuda:0"


def close(got, ref, tol=1e-4, name=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item() / scale
    assert err <= tol, "%s: max|err|/max|ref| = %.3e > %.1e" % (name, err, tol)
    return err


def nhwc(x):   # NCHW cpu -> NHWC gpu
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(x):   # NHWC gpu -> NCHW cpu
    return x.cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (45, 70, 33), (130, 992, 608), (1, 5, 3), (1440, 976, 976), (976, 976, 1440),
                                   (64, 64, 1028), (70, 33, 640), (132, 100, 612), (4, 4, 4), (68, 36, 36), (256, 192, 96), (2100, 2048, 72)])
def test_sgemm(ta, tb, M, N, K):
    g = torch.Generator().manual_seed(M * 131 + N * 7 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = ((A.t() if ta else A).double() @ (B.t() if tb else B).double() + bias.double()).float()
    C = torch.full((M, N + 3), 7.0, device=DEV)
    call("cova_sgemm", ta, tb, M, N, K, A.to(DEV), A.shape[1], B.to(DEV), B.shape[1], C, N + 3, bias.to(DEV), 0)
    close(C[:, :N], ref, 2e-5, "sgemm")
    assert (C[:, N:] == 7.0).all()
    C2 = torch.full((M, N + 3), 7.0, device=DEV)
    call("cova_sgemm", ta, tb, M, N, K, A.to(DEV), A.shape[1], B.to(DEV), B.shape[1], C2, N + 3, bias.to(DEV), 0)
    assert torch.equal(C2, C)                                  # deterministic
    call("cova_sgemm", ta, tb, M, N, K, A.to(DEV), A.shape[1], B.to(DEV), B.shape[1], C, N + 3, None, 1)
    close(C[:, :N], 2 * ref - bias, 2e-5, "sgemm accumulate")


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(1440, 992, 992), (1440, 768, 608), (768, 608, 1440), (132, 100, 612), (64, 64, 16), (2100, 2048, 72)])
def test_sgemm_lds_dma_kernel_against_the_register_staged_one(ta, tb, M, N, K):
    """cova_sgemm's two kernels (operand tiles by global -> LDS copies, taken whenever 16-byte pieces are possible; operand tiles
    staged through registers, any shape) sum a dot product in different orders: same result to f32 round-off, each of them
    bit-reproducible, sub-matrix views (leading dimensions, offset bases) included.  The shapes take all three tile forms of the
    first kernel (64 x 64 tiles with one or two k-groups per block, 32 x 64 tiles with two)."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    lda, ldb = (M if ta else K) + 8, (K if tb else N) + 4
    A = torch.randn((K if ta else M), lda, generator=g).to(DEV)
    B = torch.randn((N if tb else K), ldb, generator=g).to(DEV)
    Av, Bv = A[:, 4:], B[:, 4:]                        # 16-byte aligned views inside wider rows
    ref = ((Av[:, :M].t() if ta else Av[:, :K]).double() @ (Bv[:, :K].t() if tb else Bv[:, :N]).double()).float()
    out = []
    try:
        for v in (1, 0, 1):
            assert query("cova_set_option", 22, v) == 0
            C = torch.full((M, N + 4), 7.0, device=DEV)
            call("cova_sgemm", ta, tb, M, N, K, Av, lda, Bv, ldb, C, N + 4, None, 0)
            assert (C[:, N:] == 7.0).all()
            close(C[:, :N], ref, 2e-5, "sgemm option 22 = %d" % v)
            out.append(C)
    finally:
        query("cova_set_option", 22, 1)
    assert torch.equal(out[0], out[2])


@pytest.mark.parametrize("B,H,W", [(1, 16, 64), (2, 37, 50), (1, 64, 64)])
def test_conv1_fwd_wgrad(B, H, W):
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    wk = torch.empty(154, 64, device=DEV)
    call("cova_conv1_prep_weights", w.to(DEV), wk)
    H1, W1 = query("cova_conv_out_size", H, 7, 2, 3), query("cova_conv_out_size", W, 7, 2, 3)
    nt = query("cova_conv1_num_partials", B, H, W)
    out, part = torch.empty(B, H1, W1, 64, device=DEV), torch.empty(nt, 2, 64, device=DEV)
    call("cova_conv1_fwd", x.to(DEV), wk, out, part, B, H, W)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(x, wr, stride=2, padding=3)
    close(nchw(out), ref, 1e-4, "conv1 fwd")
    close(part[:, 0].sum(0), ref.sum((0, 2, 3)), 1e-4, "conv1 stat sum")
    close(part[:, 1].sum(0), (ref * ref).sum((0, 2, 3)), 1e-4, "conv1 stat sumsq")
    dy = torch.randn(B, 64, H1, W1, generator=g)
    (ref * dy).sum().backward()
    ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=DEV)
    dw = torch.empty(64, 3, 7, 7, device=DEV)
    call("cova_conv1_wgrad", x.to(DEV), nhwc(dy), dw, ws, B, H, W)
    close(dw, wr.grad, 2e-4, "conv1 wgrad")


@pytest.mark.parametrize("R,C,relu,res", [(300, 64, 1, 1), (77, 32, 1, 0), (129, 992, 1, 0), (50, 6, 0, 0)])
def test_batchnorm_train_fwd_bwd(R, C, relu, res):
    g = torch.Generator().manual_seed(R + C)
    x = (torch.randn(R, C, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    resid = torch.randn(R, C, generator=g) if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    if res:
        y = y + resid
    if relu:
        y = F.relu(y)
    dout = torch.randn(R, C, generator=g)
    (y * dout).sum().backward()
    params = {"bn.weight": gamma.detach().to(DEV), "bn.bias": beta.detach().to(DEV)}
    buffers = {"bn.running_mean": rm.to(DEV), "bn.running_var": rv.to(DEV),
               "bn.num_batches_tracked": torch.zeros((), dtype=torch.long, device=DEV)}
    xg = x.detach().to(DEV)
    part, n = engine.colstats(xg, C, R, C)
    st = engine.bn_params("bn.", params, buffers, C, xg, True, part, n, R)
    out = torch.empty(R, C, device=DEV)
    call("cova_bn_act_fwd", xg, C, st.scale, st.shift, resid.to(DEV) if res else None, C, out, C, R, C, relu)
    close(out, y, 1e-5, "bn fwd")
    close(buffers["bn.running_mean"], rm_ref, 1e-5, "running_mean")
    close(buffers["bn.running_var"], rv_ref, 1e-5, "running_var")
    assert int(buffers["bn.num_batches_tracked"]) == 1
    dz, dres = torch.empty(R, C, device=DEV), torch.empty(R, C, device=DEV)
    dg, db = engine.bn_backward(dout.to(DEV), C, out if relu else None, C, xg, C, st, R, dz, C, dres, C)
    close(dz, x.grad, 1e-4, "bn dz")
    close(dg, gamma.grad, 1e-4, "bn dgamma")
    close(db, beta.grad, 1e-4, "bn dbeta")
    # eval-mode parameters
    st2 = engine.bn_params("bn.", params, buffers, C, xg, False)
    call("cova_bn_act_fwd", xg, C, st2.scale, st2.shift, None, 0, out, C, R, C, 0)
    close(out, F.batch_norm(x.detach(), rm_ref, rv_ref, gamma.detach(), beta.detach(), False, 0.1, 1e-5),
          1e-5, "bn eval")


@pytest.mark.parametrize("R,C,relu,drop,pad", [(300, 64, 1, 1, 5), (77, 16, 1, 0, 5), (1440, 976, 1, 1, 5), (50, 6, 0, 0, 5),
                                               (2100, 33, 1, 1, 5),
                                               # 16-byte aligned rows: the float4 kernels (3 / 6 / 8 rows per thread; a
                                               # ragged last column block; no ReLU / no Dropout forms)
                                               (1440, 976, 1, 1, 8), (300, 64, 1, 1, 4), (77, 16, 1, 0, 0), (1440, 32, 1, 0, 8),
                                               (2880, 100, 1, 1, 4), (3500, 36, 0, 1, 0), (1537, 12, 0, 0, 4)])
def test_batchnorm1d_single_launch_fwd_bwd(R, C, relu, drop, pad):
    """cova_bn1d_fwd / cova_bn1d_bwd (statistics + finalize + apply, with the Dropout behind the layer and the column sums
    of dz riding along) against torch autograd on the CPU, and against the separate-launch path they replace."""
    g = torch.Generator().manual_seed(R + 3 * C)
    x = (torch.randn(R, C, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    p = 0.3
    keep = (torch.rand(R, C, generator=g) >= p)
    y = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    if relu:
        y = F.relu(y)
    yd = y * keep / (1 - p) if drop else y
    dout = torch.randn(R, C, generator=g)
    (yd * dout).sum().backward()
    params = {"bn.weight": gamma.detach().to(DEV), "bn.bias": beta.detach().to(DEV)}
    buffers = {"bn.running_mean": rm.to(DEV), "bn.running_var": rv.to(DEV),
               "bn.num_batches_tracked": torch.zeros((), dtype=torch.long, device=DEV)}
    xg = x.detach().to(DEV)
    out = torch.full((R, C + pad), 9.0, device=DEV)            # ld > C: the layer writes into a wider matrix
    mask = keep.to(torch.uint8).to(DEV)
    assert engine.bn1d_fused(True)
    if drop:
        st, dropped, m = engine.bn1d_fwd(xg, C, R, C, "bn.", params, buffers, True, out, C + pad, relu, drop=(p, 0, mask))
        close(dropped, yd, 1e-5, "bn1d dropped")
        assert m is mask
    else:
        st = engine.bn1d_fwd(xg, C, R, C, "bn.", params, buffers, True, out, C + pad, relu)
    close(out[:, :C], y, 1e-5, "bn1d fwd")
    assert pad == 0 or (out[:, C:] == 9.0).all()
    close(buffers["bn.running_mean"], rm_ref, 1e-5, "running_mean")
    close(buffers["bn.running_var"], rv_ref, 1e-5, "running_var")
    assert int(buffers["bn.num_batches_tracked"]) == 1
    # against the three launches it replaces: same fp64 finalize, statistics summed in another order
    part, n = engine.colstats(xg, C, R, C)
    b2 = {k: (rm.to(DEV) if "mean" in k else rv.to(DEV) if "var" in k else torch.zeros((), dtype=torch.long, device=DEV))
          for k in buffers}
    st2 = engine.bn_params("bn.", params, b2, C, xg, True, part, n, R)
    close(st.scale, st2.scale, 2e-6, "scale")
    close(st.mean, st2.mean, 2e-6, "mean")
    dz = torch.empty(R, C, device=DEV)
    colsum = torch.empty(C, device=DEV)
    dg, db = engine.bn_backward(dout.to(DEV), C, out if relu else None, C + pad, xg, C, st, R, dz, C,
                                drop=(mask, p) if drop else None, colsum=colsum)
    close(dz, x.grad, 1e-4, "bn1d dz")
    close(dg, gamma.grad, 1e-4, "bn1d dgamma")
    close(db, beta.grad, 1e-4, "bn1d dbeta")
    assert float((colsum - dz.sum(0)).abs().max()) <= 1e-4 * max(float(dz.abs().max()), 1.0) * (R ** 0.5)
    # generated mask: the same stream as cova_dropout_fwd
    if drop:
        gen = torch.empty(R, C, dtype=torch.uint8, device=DEV)
        o2, d2 = torch.empty(R, C, device=DEV), torch.empty(R, C, device=DEV)
        call("cova_bn1d_fwd", xg, C, R, C, params["bn.weight"], params["bn.bias"], None, None, None, 0.1, 1e-5, relu, o2, C,
             d2, C, gen, p, 1234, 0, st.scale, st.shift, st.mean, st.invstd)
        ref_o, ref_m = engine.dropout_fwd(o2, C, R, C, p, 1234)
        assert torch.equal(gen, ref_m) and torch.equal(d2, ref_o)
    # the two kernel forms (cova_set_option(14, .): float4 / register-resident rows against 128 row slices) agree to the
    # association of their fp64 sums
    if C % 4 == 0 and pad % 4 == 0:
        res = {}
        for variant in (0, 1):
            query("cova_set_option", 14, variant)
            try:
                o3, d3 = torch.empty(R, C + pad, device=DEV), torch.empty(R, C, device=DEV)
                stv = [torch.empty(C, device=DEV) for _ in range(4)]
                call("cova_bn1d_fwd", xg, C, R, C, params["bn.weight"], params["bn.bias"], None, None, None, 0.1, 1e-5, relu,
                     o3, C + pad, d3 if drop else None, C, mask if drop else None, p, 0, 1, *stv)
                dz3, cs3, dg3, db3 = (torch.empty(R, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV),
                                      torch.empty(C, device=DEV))
                call("cova_bn1d_bwd", dout.to(DEV), C, mask if drop else None, p, o3 if relu else None, C + pad, xg, C, stv[2],
                     stv[3], stv[0], R, C, dg3, db3, dz3, C, cs3)
                res[variant] = (o3[:, :C].clone(), d3, stv[0], stv[1], dz3, cs3, dg3, db3)
            finally:
                query("cova_set_option", 14, 1)
        for a, b, nm in zip(res[0], res[1], ("out", "dropped", "scale", "shift", "dz", "colsum", "dgamma", "dbeta")):
            if nm == "dropped" and not drop:
                continue
            if nm == "colsum":      # the column sums of dz are zero up to rounding (BatchNorm backward): an absolute bound
                assert float((b - a).abs().max()) <= 1e-4 * max(float(res[0][4].abs().max()), 1.0) * (R ** 0.5)
                continue
            close(b, a, 2e-6, "bn1d forms: " + nm)


@pytest.mark.parametrize("B,H1,W1", [(1, 8, 8), (2, 13, 21), (1, 32, 32), (2, 70, 46)])
def test_bn_relu_maxpool(B, H1, W1):
    g = torch.Generator().manual_seed(H1 * 7 + W1)
    y = torch.randn(B, 64, H1, W1, generator=g).requires_grad_(True)
    gamma = (torch.rand(64, generator=g) - 0.3).requires_grad_(True)     # some negative scales
    beta = (torch.randn(64, generator=g) * 0.2).requires_grad_(True)
    bn = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    ref, ref_idx = F.max_pool2d(F.relu(bn), 3, 2, 1, return_indices=True)
    dp = torch.randn(ref.shape, generator=g)
    (ref * dp).sum().backward()
    H2, W2 = ref.shape[2], ref.shape[3]
    yg = nhwc(y.detach())
    part, n = engine.colstats(yg, 64, B * H1 * W1, 64)
    params = {"bn.weight": gamma.detach().to(DEV), "bn.bias": beta.detach().to(DEV)}
    buffers = {"bn.running_mean": torch.zeros(64, device=DEV), "bn.running_var": torch.ones(64, device=DEV),
               "bn.num_batches_tracked": torch.zeros((), dtype=torch.long, device=DEV)}
    st = engine.bn_params("bn.", params, buffers, 64, yg, True, part, n, B * H1 * W1)
    out = torch.empty(B, H2, W2, 64, device=DEV)
    idx = torch.empty(B, H2, W2, 64, device=DEV, dtype=torch.uint8)
    ymax = torch.empty_like(out)
    call("cova_bn_relu_maxpool_fwd", yg, st.scale, st.shift, out, idx, ymax, B, H1, W1)
    close(nchw(out), ref, 1e-5, "maxpool fwd")
    close(torch.relu(ymax * st.scale + st.shift), out, 1e-6, "arg-max pre-activation")
    npart = query("cova_bn_relu_maxpool_bwd_num_partials", B, H1, W1)
    bpart = torch.empty(npart, 2, 64, device=DEV)
    dpg = nhwc(dp)
    call("cova_bn_relu_maxpool_bwd_reduce", dpg, idx, yg, st.scale, st.shift, st.mean, st.invstd, bpart, B, H1, W1)
    dg, db, coef = (torch.empty(64, device=DEV), torch.empty(64, device=DEV), torch.empty(2, 64, device=DEV))
    call("cova_bn_finalize_bwd", bpart, npart, 64, float(B * H1 * W1), dg, db, coef)
    dz = torch.empty(B, H1, W1, 64, device=DEV)
    call("cova_bn_relu_maxpool_bwd_apply", dpg, idx, yg, st.scale, st.shift, st.mean, st.invstd, coef, dz, B, H1, W1)
    close(dg, gamma.grad, 1e-4, "pool dgamma")
    close(db, beta.grad, 1e-4, "pool dbeta")
    close(nchw(dz), y.grad, 1e-4, "pool dz")


def test_roipool_matches_oracle_bit_exact():
    rs = np.random.RandomState(2)
    B, C, H, W = 3, 64, 20, 24
    feat = torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32))
    n = 150
    x1 = rs.uniform(-8, 90, n); y1 = rs.uniform(-8, 75, n)
    rois = np.stack([rs.randint(0, B, n), x1, y1, x1 + rs.uniform(0.1, 60, n), y1 + rs.uniform(0.1, 50, n)], 1)
    rois[:5, 1:] = [[200, 200, 300, 300]] * 5                     # outside the map -> empty bins
    rois[5:8, 1:] = np.round(rois[5:8, 1:] / 4) * 4 + 2.0         # x.5 after scaling: round-half cases
    rois = torch.from_numpy(rois.astype(np.float32))
    ref, ref_arg = O.roi_pool_argmax(feat, rois, (3, 3), 0.25)
    out = torch.full((n, 600), -5.0, device=DEV)
    arg = torch.empty((n, 576), device=DEV, dtype=torch.int32)
    call("cova_roipool_fwd", nhwc(feat), rois.to(DEV), n, B, C, H, W, 3, 3, 0.25, out, 600, arg)
    assert torch.equal(out[:, :576].cpu(), ref.reshape(n, 576))
    assert torch.equal(arg.cpu(), ref_arg.reshape(n, 576))
    assert (out[:, 576:] == -5.0).all()
    gout = torch.from_numpy(rs.standard_normal((n, 576)).astype(np.float32))
    gref = torch.empty(B, C, H, W)
    import ctypes
    O._lib().oracle_roipool_bwd(O._fp(gout), O._fp(rois), O._fp(ref_arg), n, B, C, H, W, 3, 3, O._fp(gref))
    gfeat = torch.empty(B, H, W, C, device=DEV)
    call("cova_roipool_bwd", gout.to(DEV), 576, rois.to(DEV), arg, n, B, C, H, W, 3, 3, 0.25, gfeat, torch.empty(query("cova_roipool_bwd_workspace_words", n, B, C, 3, 3), dtype=torch.int32, device=DEV))
    close(nchw(gfeat), gref, 1e-5, "roipool bwd")
    # un-materialised feature map: feat = relu(scale*z + shift + x) formed inside the forward kernel, which
    # also keeps z at every arg-max; the backward masks by pooled > 0 and takes the BatchNorm-backward sums per
    # pooled entry -- no map is read
    z = nhwc(torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)))
    xres = nhwc(torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)))
    scale = torch.from_numpy(rs.uniform(-0.5, 1.5, C).astype(np.float32)).to(DEV)
    shift = torch.from_numpy(rs.standard_normal(C).astype(np.float32)).to(DEV) * 0.3
    mean = torch.from_numpy(rs.standard_normal(C).astype(np.float32)).to(DEV) * 0.2
    invstd = torch.from_numpy(rs.uniform(0.5, 1.5, C).astype(np.float32)).to(DEV)
    fmat = torch.empty(B, H, W, C, device=DEV)
    call("cova_bn_act_fwd", z, C, scale, shift, xres, C, fmat, C, B * H * W, C, 1)
    out_m, arg_m = torch.empty(n, 576, device=DEV), torch.empty((n, 576), device=DEV, dtype=torch.int32)
    out_l, arg_l = torch.empty(n, 576, device=DEV), torch.empty((n, 576), device=DEV, dtype=torch.int32)
    zmax = torch.empty(n, 576, device=DEV)
    call("cova_roipool_fwd", fmat, rois.to(DEV), n, B, C, H, W, 3, 3, 0.25, out_m, 576, arg_m)
    call("cova_roipool_fwd_bn", z, xres, scale, shift, rois.to(DEV), n, B, C, H, W, 3, 3, 0.25, out_l, 576, arg_l, zmax)
    assert torch.equal(out_l, out_m) and torch.equal(arg_l, arg_m)
    valid = arg_l >= 0
    pos = arg_l.clamp_min(0).long()
    page = rois[:, 0].long().to(DEV).view(n, 1)
    chan = (torch.arange(576, device=DEV) // 9).view(1, 576)
    assert torch.equal(zmax[valid], z.reshape(B, H * W, C)[page, pos, chan][valid])
    scratch = torch.empty(query("cova_roipool_bwd_workspace_words", n, B, C, 3, 3), dtype=torch.int32, device=DEV)
    g_plain = torch.empty(B, H, W, C, device=DEV)
    call("cova_roipool_bwd", gout.to(DEV), 576, rois.to(DEV), arg_l, n, B, C, H, W, 3, 3, 0.25, g_plain, scratch)
    npart = query("cova_roipool_bwd_bn_num_partials", n)
    part = torch.empty(npart, 2, C, device=DEV)
    gmask = torch.empty(B, H, W, C, device=DEV)
    call("cova_roipool_bwd_bn", gout.to(DEV), 576, out_l, 576, zmax, rois.to(DEV), arg_l, n, B, C, H, W, 3, 3, 0.25,
         mean, invstd, gmask, part, scratch)
    ref_masked = g_plain * (fmat > 0)
    close(gmask, ref_masked, 1e-5, "roipool bwd masked")
    close(part[:, 0].sum(0), ref_masked.sum((0, 1, 2)), 1e-4, "roipool bwd sum dy")
    close(part[:, 1].sum(0), (ref_masked * ((z - mean) * invstd)).sum((0, 1, 2)), 1e-4, "roipool bwd sum dy*xhat")
    # the BatchNorm-backward finalize as the entry pass's tail: bit-identical to the separate launch
    st = engine.bn_state(C, z, B * H * W, True)
    st.mean.copy_(mean), st.invstd.copy_(invstd), st.scale.copy_(scale)
    dg_r, db_r, abc_r = engine._bn_abc_from_partials(part, npart, st, B * H * W, None, "x.", z)
    dg_t, db_t, abc_t, tail = engine.bn_tail_bwd(st, B * H * W, None, "x.", z)
    part_t, gmask_t = torch.empty_like(part), torch.empty_like(gmask)
    call("cova_roipool_bwd_bn_tail", gout.to(DEV), 576, out_l, 576, zmax, rois.to(DEV), arg_l, n, B, C, H, W, 3, 3, 0.25,
         mean, invstd, gmask_t, part_t, scratch, tail.ptr)
    assert torch.equal(gmask_t, gmask) and torch.equal(part_t, part)
    assert torch.equal(abc_t, abc_r) and torch.equal(dg_t, dg_r) and torch.equal(db_t, db_r)
    # memory safety: page indices outside [0, B) pool nothing and scatter nothing
    bad = rois.clone()
    bad[::7, 0] = float(B + 3)
    bad[3::11, 0] = -2.0
    isbad = ((bad[:, 0] < 0) | (bad[:, 0] >= B))
    out_b, arg_b = torch.empty(n, 576, device=DEV), torch.empty((n, 576), device=DEV, dtype=torch.int32)
    call("cova_roipool_fwd", nhwc(feat), bad.to(DEV), n, B, C, H, W, 3, 3, 0.25, out_b, 576, arg_b)
    assert (out_b.cpu()[isbad] == 0).all() and (arg_b.cpu()[isbad] == -1).all()
    assert torch.equal(out_b.cpu()[~isbad], ref.reshape(n, 576)[~isbad])
    g_b = torch.empty(B, H, W, C, device=DEV)
    call("cova_roipool_bwd", gout.to(DEV), 576, bad.to(DEV), arg_b, n, B, C, H, W, 3, 3, 0.25, g_b, torch.empty(query("cova_roipool_bwd_workspace_words", n, B, C, 3, 3), dtype=torch.int32, device=DEV))
    assert torch.isfinite(g_b).all()


def test_bbox_linear():
    rs = np.random.RandomState(4)
    n, hd = 57, 32
    bb = torch.from_numpy(np.concatenate([np.zeros((n, 1)), rs.uniform(0, 1000, (n, 2)),
                                          rs.uniform(1000, 1280, (n, 2))], 1).astype(np.float32))
    W = torch.from_numpy(rs.uniform(-0.4, 0.4, (hd, 5)).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rs.uniform(-0.4, 0.4, hd).astype(np.float32)).requires_grad_(True)
    raw_ref = O.bbox_features_raw(bb)
    z_ref = F.linear(raw_ref, W, b)
    raw, z = torch.empty(n, 5, device=DEV), torch.empty(n, hd, device=DEV)
    call("cova_bbox_linear_fwd", bb.to(DEV), W.detach().to(DEV), b.detach().to(DEV), raw, z, n, hd)
    assert torch.equal(raw.cpu(), raw_ref)
    close(z, z_ref, 1e-5, "bbox z")
    dz = torch.from_numpy(rs.standard_normal((n, hd)).astype(np.float32))
    (z_ref * dz).sum().backward()
    dW, db = torch.empty(hd, 5, device=DEV), torch.empty(hd, device=DEV)
    call("cova_bbox_linear_bwd", dz.to(DEV), raw, dW, db, n, hd)
    close(dW, W.grad, 1e-4, "bbox dW")
    close(db, b.grad, 1e-4, "bbox db")


def test_head_kernels():
    rs = np.random.RandomState(9)
    n, T, nc = 203, 992, 4
    x = torch.from_numpy(rs.standard_normal((n, T)).astype(np.float32)).requires_grad_(True)
    W = torch.from_numpy(rs.uniform(-0.1, 0.1, (nc, T)).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rs.uniform(-0.1, 0.1, nc).astype(np.float32)).requires_grad_(True)
    labels = torch.from_numpy(rs.randint(0, nc, n))
    logits_ref = F.linear(x, W, b)
    loss_ref = F.cross_entropy(logits_ref, labels, reduction="sum")
    loss_ref.backward()
    logits = torch.empty(n, nc, device=DEV)
    call("cova_linear_small_fwd", x.detach().to(DEV), T, W.detach().to(DEV), b.detach().to(DEV), logits, n, T, nc)
    close(logits, logits_ref, 1e-5, "linear_small fwd")
    loss, dl, pred = engine.ce_sum(logits, labels.to(DEV))
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    assert torch.equal(pred.cpu(), logits.cpu().argmax(1))
    dx, dW, db = torch.empty(n, T, device=DEV), torch.empty(nc, T, device=DEV), torch.empty(nc, device=DEV)
    call("cova_linear_small_bwd", dl, x.detach().to(DEV), T, W.detach().to(DEV), dx, T, dW, db, n, T, nc)
    close(dx, x.grad, 1e-4, "linear_small dx")
    close(dW, W.grad, 1e-4, "linear_small dW")
    close(db, b.grad, 1e-4, "linear_small db")
    cs = torch.empty(T, device=DEV)
    call("cova_colsum", x.detach().to(DEV), T, n, T, cs)
    close(cs, x.detach().sum(0), 1e-5, "colsum")


def test_dropout():
    n, T, p = 311, 992, 0.2
    x = torch.randn(n, T, device=DEV)
    out, mask = engine.dropout_fwd(x, T, n, T, p, 1234)
    out2, mask2 = engine.dropout_fwd(x, T, n, T, p, 1234)
    assert torch.equal(mask, mask2) and torch.equal(out, out2)          # same seed -> same mask
    _, mask3 = engine.dropout_fwd(x, T, n, T, p, 1235)
    assert not torch.equal(mask, mask3)
    keep = mask.float().mean().item()
    assert abs(keep - (1 - p)) < 0.01
    assert torch.allclose(out, x * mask.float() / (1 - p), rtol=1e-6, atol=0)
    given = (torch.rand(n, T, device=DEV) > 0.5).to(torch.uint8)
    out4, _ = engine.dropout_fwd(x, T, n, T, p, 0, given)
    assert torch.allclose(out4, x * given.float() / (1 - p), rtol=1e-6, atol=0)
    dx = torch.empty(n, T, device=DEV)
    call("cova_dropout_bwd", x, T, given, dx, T, n, T, p)
    assert torch.equal(dx, out4)


@pytest.mark.parametrize("M,N,K", [(311, 992, 992), (45, 70, 33), (1440, 976, 976)])
def test_sgemm_with_dropout_backward_epilogue(M, N, K):
    """cova_sgemm_dropout_bwd = cova_sgemm followed by cova_dropout_bwd, bit for bit (decoder_bwd's dL/d(comb))."""
    g = torch.Generator().manual_seed(M + N)
    A, B = torch.randn(M, K, generator=g).to(DEV), torch.randn(K, N, generator=g).to(DEV)
    keep = (torch.rand(M, N, generator=g) > 0.2).to(torch.uint8).to(DEV)
    tmp, ref, got = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV), torch.full((M, N + 4), 7.0, device=DEV)
    call("cova_sgemm", 0, 0, M, N, K, A, K, B, N, tmp, N, None, 0)
    call("cova_dropout_bwd", tmp, N, keep, ref, N, M, N, 0.2)
    call("cova_sgemm_dropout_bwd", 0, 0, M, N, K, A, K, B, N, got, N + 4, keep, 0.2)
    assert torch.equal(got[:, :N], ref) and (got[:, N:] == 7.0).all()


def test_adam_matches_torch():
    rs = np.random.RandomState(1)
    n = 10007
    p0 = torch.from_numpy(rs.standard_normal(n).astype(np.float32))
    ref_p, state = [p0.clone()], None
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gr = torch.from_numpy(rs.standard_normal(n).astype(np.float32))
        ref_p, state = O.adam_reference(ref_p, [gr], state)
        call("cova_adam_step", p, gr.to(DEV), m, v, n, step, 5e-4, 0.9, 0.999, 1e-8, 1e-3)
        close(p, ref_p[0], 1e-6, "adam step %d" % step)


def test_conv1_variants_multi_tile():
    B, H, W = 2, 150, 330                     # 10 x 6 x 2 = 120 tiles; capped grid -> many per block
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, 3, H, W, generator=g)
    wr = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).requires_grad_(True)
    ref = F.conv2d(x, wr, stride=2, padding=3)
    H1, W1 = ref.shape[2], ref.shape[3]
    dy = torch.randn(B, 64, H1, W1, generator=g)
    (ref * dy).sum().backward()
    wk = torch.empty(154, 64, device=DEV)
    call("cova_conv1_prep_weights", wr.detach().to(DEV), wk)
    nt = query("cova_conv1_num_partials", B, H, W)
    ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=DEV)
    for cap in (0, 7):
        query("cova_set_option", 2, cap)
        nt = query("cova_conv1_num_partials", B, H, W)        # depends on the grid cap
        out, part = torch.zeros(B, H1, W1, 64, device=DEV), torch.zeros(nt, 2, 64, device=DEV)
        call("cova_conv1_fwd", x.to(DEV), wk, out, part, B, H, W)
        close(nchw(out), ref, 1e-4, "conv1 fwd cap %d" % cap)
        close(part[:, 1].sum(0), (ref * ref).sum((0, 2, 3)), 1e-4, "conv1 sumsq")
        dw = torch.zeros(64, 3, 7, 7, device=DEV)
        call("cova_conv1_wgrad", x.to(DEV), nhwc(dy), dw, ws, B, H, W)
        close(dw, wr.grad, 2e-4, "conv1 wgrad cap %d" % cap)
    query("cova_set_option", 2, 0)


def test_conv3x3_dgrad_bnbwd_fusion_matches_unfused():
    """Fused data-gradient epilogue (ReLU mask + BatchNorm-backward sums) == plain data gradient followed by
    cova_bn_bwd_reduce, on a multi-tile problem, for both Winograd kernel families."""
    B, H, W = 2, 37, 70
    g = torch.Generator().manual_seed(11)
    dz = nhwc(torch.randn(B, 64, H, W, generator=g))
    add = nhwc(torch.randn(B, 64, H, W, generator=g))
    act = nhwc(torch.relu(torch.randn(B, 64, H, W, generator=g)))
    z = nhwc(torch.randn(B, 64, H, W, generator=g) * 2 + 0.3)
    mean = (torch.randn(64, generator=g) * 0.2).to(DEV)
    invstd = (torch.rand(64, generator=g) + 0.5).to(DEV)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    u2f, u2d = torch.empty(16, 16, 4, 64, device=DEV), torch.empty(16, 16, 4, 64, device=DEV)
    call("cova_conv3x3_prep_weights_wino", w.to(DEV), u2f, u2d)
    u4f, u4d = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV)
    call("cova_conv3x3_wino4_prep", w.to(DEV), u4f, u4d)
    R = B * H * W
    n = query("cova_colreduce_num_chunks", R, 64)
    for fam in ("w2", "w4"):
        plain = torch.empty(B, H, W, 64, device=DEV)
        if fam == "w2":
            nt = query("cova_conv3x3_wino_num_partials", B, H, W)
            call("cova_conv3x3_wino", dz, u2d, add, None, None, None, None, plain, None, B, H, W)
        else:
            nt = query("cova_conv3x3_wino4_num_partials", B, H, W)
            call("cova_conv3x3_wino4_full", dz, None, None, 0, u4d, add, None, None, None, None, None, None, plain, None, B, H, W)
        part_ref = torch.empty(n, 2, 64, device=DEV)
        call("cova_bn_bwd_reduce", plain, 64, act, 64, z, 64, mean, invstd, R, 64, part_ref)
        dy, part = torch.empty(B, H, W, 64, device=DEV), torch.empty(nt, 2, 64, device=DEV)
        if fam == "w2":
            call("cova_conv3x3_wino", dz, u2d, add, act, z, mean, invstd, dy, part, B, H, W)
        else:
            call("cova_conv3x3_wino4_full", dz, None, None, 0, u4d, add, act, None, None, z, mean, invstd, dy, part, B, H, W)
        assert torch.equal(dy, plain * (act > 0)), fam
        close(part.sum(0), part_ref.sum(0), 1e-5, "fused bn-bwd sums " + fam)


@pytest.mark.parametrize("geo", [1, 2])
@pytest.mark.parametrize("B,H,W,cap", [(1, 8, 32, 0), (2, 19, 45, 0), (3, 100, 200, 0), (3, 100, 200, 5)])
def test_conv3x3_winograd_matches_cpu(B, H, W, cap, geo):
    """Winograd F(2x2,3x3) kernel: forward (+statistics), data gradient (+addend) and the fused
    ReLU-mask / BN-backward epilogue against torch-CPU."""
    g = torch.Generator().manual_seed(H * W + B)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    dz = torch.randn(B, 64, H, W, generator=g)
    add = torch.randn(B, 64, H, W, generator=g)
    uf, ud = torch.empty(16, 16, 4, 64, device=DEV), torch.empty(16, 16, 4, 64, device=DEV)
    call("cova_conv3x3_prep_weights_wino", w.to(DEV), uf, ud)
    query("cova_set_option", 2, cap)
    query("cova_set_option", 6, geo)          # 8x32 tiles / 1 block per CU | 8x16 tiles / 2 blocks per CU
    try:
        ntw = query("cova_conv3x3_wino_num_partials", B, H, W)
        out, part = torch.zeros(B, H, W, 64, device=DEV), torch.zeros(ntw, 2, 64, device=DEV)
        call("cova_conv3x3_wino", nhwc(x), uf, None, None, None, None, None, out, part, B, H, W)
        ref = F.conv2d(x, w, padding=1)
        close(nchw(out), ref, 1e-4, "winograd fwd")
        close(part[:, 0].sum(0), ref.sum((0, 2, 3)), 1e-4, "winograd stat sum")
        close(part[:, 1].sum(0), (ref * ref).sum((0, 2, 3)), 1e-4, "winograd stat sumsq")
        xr = x.clone().requires_grad_(True)
        (F.conv2d(xr, w, padding=1) * dz).sum().backward()
        dx = torch.zeros(B, H, W, 64, device=DEV)
        call("cova_conv3x3_wino", nhwc(dz), ud, nhwc(add), None, None, None, None, dx, None, B, H, W)
        close(nchw(dx), xr.grad + add, 1e-4, "winograd dgrad")
        # fused epilogue: dy = (dgrad + addend) * (act > 0), sums (sum dy, sum dy*xhat) -- against torch
        actc = torch.relu(torch.randn(B, 64, H, W, generator=g))
        zc = torch.randn(B, 64, H, W, generator=g)
        act, z = nhwc(actc), nhwc(zc)
        meanc, invstdc = torch.randn(64, generator=g) * 0.2, torch.rand(64, generator=g) + 0.5
        mean, invstd = meanc.to(DEV), invstdc.to(DEV)
        dy_ref = (xr.grad + add) * (actc > 0)
        xhat = (zc - meanc.view(1, -1, 1, 1)) * invstdc.view(1, -1, 1, 1)
        dy_w, part_w = torch.empty(B, H, W, 64, device=DEV), torch.empty(ntw, 2, 64, device=DEV)
        call("cova_conv3x3_wino", nhwc(dz), ud, nhwc(add), act, z, mean, invstd, dy_w, part_w, B, H, W)
        close(nchw(dy_w), dy_ref, 1e-4, "winograd fused dy")
        close(part_w[:, 0].sum(0), dy_ref.sum((0, 2, 3)), 1e-4, "winograd fused sum dy")
        close(part_w[:, 1].sum(0), (dy_ref * xhat).sum((0, 2, 3)), 1e-4, "winograd fused sum dy*xhat")
    finally:
        query("cova_set_option", 2, 0)
        query("cova_set_option", 6, 1)


@pytest.mark.parametrize("B,H,W,cap", [(1, 8, 32, 0), (2, 19, 45, 0), (3, 100, 200, 0), (3, 100, 200, 5)])
def test_conv3x3_winograd_wgrad(B, H, W, cap):
    g = torch.Generator().manual_seed(H + W + B)
    x = torch.randn(B, 64, H, W, generator=g)
    dz = torch.randn(B, 64, H, W, generator=g)
    wr = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).requires_grad_(True)
    (F.conv2d(x, wr, padding=1) * dz).sum().backward()
    ws = torch.empty(query("cova_conv3x3_wgrad_workspace_floats", B, H, W), device=DEV)
    query("cova_set_option", 2, cap)
    try:
        dw = torch.zeros(64, 64, 3, 3, device=DEV)
        call("cova_conv3x3_wgrad_wino", nhwc(x), nhwc(dz), dw, ws, B, H, W)
        close(dw, wr.grad, 2e-4, "winograd wgrad")
    finally:
        query("cova_set_option", 2, 0)


@pytest.mark.parametrize("geo", [1, 2])
@pytest.mark.parametrize("B,H,W,cap", [(1, 8, 32, 0), (2, 19, 45, 0), (3, 100, 200, 0), (3, 100, 200, 5)])
def test_conv3x3_winograd_affine_on_load(B, H, W, cap, geo):
    """cova_conv3x3_wino_pro == cova_conv3x3_wino on a pre-transformed input (bit-exact: the
    prologue evaluates the same fma as cova_bn_act_fwd), for BatchNorm+ReLU on load, the
    BatchNorm-backward apply on load, and the z-derived ReLU mask in the epilogue."""
    g = torch.Generator().manual_seed(7 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x, x2 = nhwc(rnd(B, 64, H, W)), nhwc(rnd(B, 64, H, W))
    add = nhwc(rnd(B, 64, H, W))
    w = rnd(64, 64, 3, 3) * 0.05
    abc = rnd(3, 64).to(DEV)
    uf, ud = torch.empty(16, 16, 4, 64, device=DEV), torch.empty(16, 16, 4, 64, device=DEV)
    call("cova_conv3x3_prep_weights_wino", w.to(DEV), uf, ud)
    query("cova_set_option", 6, geo)
    query("cova_set_option", 2, cap)
    nt = query("cova_conv3x3_wino_num_partials", B, H, W)        # rows = persistent blocks: after the options
    R = B * H * W
    try:
        # (a) relu(A*x + C) on load, with statistics
        a1 = torch.empty_like(x)
        call("cova_bn_act_fwd", x, 64, abc[0], abc[2], None, 0, a1, 64, R, 64, 1)
        ref, pref = torch.empty_like(x), torch.empty(nt, 2, 64, device=DEV)
        call("cova_conv3x3_wino", a1, uf, None, None, None, None, None, ref, pref, B, H, W)
        out, part = torch.empty_like(x), torch.empty(nt, 2, 64, device=DEV)
        call("cova_conv3x3_wino_pro", x, None, abc, 1, uf, None, None, None, None, None, None, None,
             out, part, B, H, W)
        assert torch.equal(out, ref) and torch.equal(part, pref)
        # (b) A*x + B*x2 + C on load (no relu), addend, epilogue mask from z
        pre = torch.addcmul(torch.addcmul(abc[2].expand_as(x), x2, abc[1]), x, abc[0])
        z = nhwc(rnd(B, 64, H, W))
        msc, msh = rnd(64).to(DEV), rnd(64).to(DEV) * 0.3
        act = torch.empty_like(z)
        call("cova_bn_act_fwd", z, 64, msc, msh, None, 0, act, 64, R, 64, 1)
        mean, invstd = rnd(64).to(DEV) * 0.2, (torch.rand(64, generator=g) + 0.5).to(DEV)
        call("cova_conv3x3_wino", pre, ud, add, act, z, mean, invstd, ref, pref, B, H, W)
        call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, add, None, msc, msh, z, mean, invstd, out,
             part, B, H, W)
        assert torch.equal(out == 0, ref == 0), "ReLU mask from z differs from the materialised one"
        close(out, ref, 2e-6, "affine-on-load dgrad")
        close(part.sum(0), pref.sum(0), 1e-5, "affine-on-load sums")
        # (c) no prologue through the _pro entry point, no statistics
        call("cova_conv3x3_wino", x, uf, add, None, None, None, None, ref, None, B, H, W)
        call("cova_conv3x3_wino_pro", x, None, None, 0, uf, add, None, None, None, None, None, None, out,
             None, B, H, W)
        assert torch.equal(out, ref)
    finally:
        query("cova_set_option", 2, 0)
        query("cova_set_option", 6, 1)


@pytest.mark.parametrize("B,H,W,cap", [(1, 8, 32, 0), (2, 19, 45, 0), (3, 100, 200, 0), (3, 100, 200, 5), (1, 3, 5, 0)])
def test_conv3x3_winograd_f4x4_full_form(B, H, W, cap):
    """cova_conv3x3_wino4_full (F(4x4,3x3): one- and two-tensor affine prologue, addend, ReLU mask from z or from the
    materialised activation, BatchNorm-backward sums) against the F(2x2,3x3) kernel on the same inputs -- the ReLU
    masks must be IDENTICAL (same fma as cova_bn_act_fwd), values and sums agree to F(4x4)'s fp32 error -- and against
    torch in fp64.  cap forces several tiles per persistent block (the pipeline crosses tile boundaries)."""
    g = torch.Generator().manual_seed(13 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    xc, x2c, addc = rnd(B, 64, H, W), rnd(B, 64, H, W), rnd(B, 64, H, W)
    x, x2, add = nhwc(xc), nhwc(x2c), nhwc(addc)
    w = rnd(64, 64, 3, 3) * 0.05
    abc = rnd(3, 64)
    u2f, u2d = torch.empty(16, 16, 4, 64, device=DEV), torch.empty(16, 16, 4, 64, device=DEV)
    call("cova_conv3x3_prep_weights_wino", w.to(DEV), u2f, u2d)
    u4f, u4d = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV)
    call("cova_conv3x3_wino4_prep", w.to(DEV), u4f, u4d)
    query("cova_set_option", 2, cap)
    n2, n4 = query("cova_conv3x3_wino_num_partials", B, H, W), query("cova_conv3x3_wino4_num_partials", B, H, W)
    R = B * H * W
    abcd = abc.to(DEV)
    try:
        z = nhwc(rnd(B, 64, H, W))
        msc, msh = rnd(64).to(DEV), rnd(64).to(DEV) * 0.3
        act = torch.empty_like(z)
        call("cova_bn_act_fwd", z, 64, msc, msh, None, 0, act, 64, R, 64, 1)
        mean, invstd = rnd(64).to(DEV) * 0.2, (torch.rand(64, generator=g) + 0.5).to(DEV)
        pre = abc[0].view(1, -1, 1, 1) * xc + abc[1].view(1, -1, 1, 1) * x2c + abc[2].view(1, -1, 1, 1)
        dref = F.conv2d(pre.double(), w.double().flip(2, 3).transpose(0, 1), padding=1)       # data-gradient form
        cases = [  # (in2, abc, relu, weights 2/4, addend, act, msc/msh, z)
            ("two-tensor prologue, mask from z", x2, abcd, 0, (u2d, u4d), None, None, (msc, msh), z),
            ("two-tensor prologue, addend, mask from z", x2, abcd, 0, (u2d, u4d), add, None, (msc, msh), z),
            ("two-tensor prologue, addend, mask from act", x2, abcd, 0, (u2d, u4d), add, act, (None, None), z),
            ("relu prologue, statistics", None, abcd, 1, (u2f, u4f), None, None, (None, None), None),
            ("plain, addend, no statistics", None, None, 0, (u2f, u4f), add, None, (None, None), None),
        ]
        for name, in2, pabc, relu, (u2, u4), ad, ac, (sc, sh), zz in cases:
            want_part = zz is not None or "statistics" in name and "no statistics" not in name
            ref, out = torch.empty_like(x), torch.full_like(x, 7.0)
            pref = torch.empty(n2, 2, 64, device=DEV) if want_part else None
            part = torch.empty(n4, 2, 64, device=DEV) if want_part else None
            mm = (mean, invstd) if zz is not None else (None, None)
            call("cova_conv3x3_wino_pro", x, in2, pabc, relu, u2, ad, ac, sc, sh, zz, mm[0], mm[1], ref, pref, B, H, W)
            call("cova_conv3x3_wino4_full", x, in2, pabc, relu, u4, ad, ac, sc, sh, zz, mm[0], mm[1], out, part, B, H, W)
            if zz is not None:
                assert torch.equal(out == 0, ref == 0), name + ": ReLU masks differ"
            close(out, ref, 2e-5, name)
            if want_part:
                close(part.double().sum(0), pref.double().sum(0), 2e-4, name + " sums")
        # against torch (fp64): the two-tensor prologue + addend with the mask from z
        out, part = torch.empty_like(x), torch.empty(n4, 2, 64, device=DEV)
        call("cova_conv3x3_wino4_full", x, x2, abcd, 0, u4d, add, None, msc, msh, z, mean, invstd, out, part, B, H, W)
        gate = (nchw(act) > 0)
        t = (dref + addc.double()) * gate
        close(nchw(out), t, 2e-5, "wino4 full vs torch")
        xhat = (nchw(z).double() - mean.cpu().double().view(1, -1, 1, 1)) * invstd.cpu().double().view(1, -1, 1, 1)
        close(part[:, 0].double().sum(0), t.sum((0, 2, 3)), 2e-4, "wino4 full sum g")
        close(part[:, 1].double().sum(0), (t * xhat).sum((0, 2, 3)), 2e-4, "wino4 full sum g*xhat")
    finally:
        query("cova_set_option", 2, 0)


@pytest.mark.parametrize("B,H,W,cap", [(1, 8, 32, 0), (2, 19, 45, 0), (3, 100, 200, 0), (3, 100, 200, 5)])
def test_batchnorm_finalize_as_launch_tail(B, H, W, cap):
    """The BatchNorm finalize riding on the producing launch (csrc/bn_tail.h: last block, ticket counter) writes exactly
    what the separate cova_bn_finalize_fwd / cova_bn_finalize_bwd_abc launches write -- bit for bit, launch after launch
    (the counter resets itself) -- for the F(4x4,3x3) forward (statistics) and data-gradient (backward sums) forms."""
    from cova_web_object_detection_amd import engine
    g = torch.Generator().manual_seed(17 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x, x2, z = nhwc(rnd(B, 64, H, W)), nhwc(rnd(B, 64, H, W)), nhwc(rnd(B, 64, H, W))
    w = rnd(64, 64, 3, 3) * 0.05
    u4f, u4d = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV)
    call("cova_conv3x3_wino4_prep", w.to(DEV), u4f, u4d)
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(DEV), rnd(64).to(DEV)
    query("cova_set_option", 2, cap)
    n4 = query("cova_conv3x3_wino4_num_partials", B, H, W)
    R = float(B * H * W)
    try:
        for rep in range(3):
            # ---- forward statistics
            rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
            nbt = torch.zeros((), dtype=torch.int64, device=DEV)
            rm2, rv2, nbt2 = rm.clone(), rv.clone(), nbt.clone()
            out, part = torch.empty_like(x), torch.empty(n4, 2, 64, device=DEV)
            call("cova_conv3x3_wino4_full", x, None, None, 0, u4f, None, None, None, None, None, None, None, out, part, B, H, W)
            ref = [torch.empty(64, device=DEV) for _ in range(4)]
            call("cova_bn_finalize_fwd", part, n4, 64, R, gamma, beta, rm, rv, nbt, 0.1, 1e-5, *ref)
            got = [torch.full((64,), 7.0, device=DEV) for _ in range(4)]
            tail = engine.BnTail(1, x, R, gamma=gamma, beta=beta, running_mean=rm2, running_var=rv2, num_batches_tracked=nbt2,
                                 scale=got[0], shift=got[1], mean=got[2], invstd=got[3])
            out2, part2 = torch.empty_like(x), torch.empty(n4, 2, 64, device=DEV)
            call("cova_conv3x3_wino4_full_tail", x, None, None, 0, u4f, None, None, None, None, None, None, None, None, out2,
                 part2, B, H, W, tail.ptr)
            assert torch.equal(out, out2) and torch.equal(part, part2)
            for a, b_ in zip(got + [rm2, rv2], ref + [rm, rv]):
                assert torch.equal(a, b_), "forward tail differs (launch %d)" % rep
            assert int(nbt2) == 1 == int(nbt)
            # ---- backward sums (two-tensor prologue, mask from z)
            abc = rnd(3, 64).to(DEV)
            msc, msh = rnd(64).to(DEV), rnd(64).to(DEV) * 0.3
            mean, invstd, scale = ref[2], ref[3], ref[0]
            call("cova_conv3x3_wino4_full", x, x2, abc, 0, u4d, None, None, msc, msh, z, mean, invstd, out, part, B, H, W)
            dg, db, abc_ref = torch.empty(64, device=DEV), torch.empty(64, device=DEV), torch.empty(3, 64, device=DEV)
            call("cova_bn_finalize_bwd_abc", part, n4, 64, R, dg, db, mean, invstd, scale, abc_ref)
            dg2, db2, abc2 = torch.empty(64, device=DEV), torch.empty(64, device=DEV), torch.empty(3, 64, device=DEV)
            tail = engine.BnTail(2, x, R, mean=mean, invstd=invstd, scale=scale, dgamma=dg2, dbeta=db2, abc=abc2)
            call("cova_conv3x3_wino4_full_tail", x, x2, abc, 0, u4d, None, None, None, msc, msh, z, mean, invstd, out2, part2,
                 B, H, W, tail.ptr)
            assert torch.equal(out, out2) and torch.equal(part, part2)
            assert torch.equal(dg, dg2) and torch.equal(db, db2) and torch.equal(abc_ref, abc2), "backward tail differs"
            # ---- mask from a materialised activation, as the map or as the bits cova_bn_act_fwd_bits leaves: identical
            act, act2, bits = torch.empty_like(x), torch.empty_like(x), torch.empty(B * H * W, 2, dtype=torch.int32, device=DEV)
            call("cova_bn_act_fwd", z, 64, msc, msh, x2, 64, act, 64, B * H * W, 64, 1)
            call("cova_bn_act_fwd_bits", z, msc, msh, x2, act2, bits, B * H * W)
            assert torch.equal(act, act2)
            sh = torch.arange(32, device=DEV, dtype=torch.int32).view(1, 1, 32)
            assert torch.equal(((bits.view(-1, 2, 1) >> sh) & 1).view(B, H, W, 64).bool(), act > 0)
            add = x2 * 0.5
            call("cova_conv3x3_wino4_full_tail", x, x2, abc, 0, u4d, add, act, None, None, None, z, mean, invstd, out, part,
                 B, H, W, tail.ptr)
            call("cova_conv3x3_wino4_full_tail", x, x2, abc, 0, u4d, add, None, bits, None, None, z, mean, invstd, out2, part2,
                 B, H, W, tail.ptr)
            assert torch.equal(out, out2) and torch.equal(part, part2)
    finally:
        query("cova_set_option", 2, 0)


def test_conv1_forward_with_batchnorm_tail():
    """cova_conv1_fwd_tail (OIHW weight read directly, BatchNorm finalize as the launch's tail) == cova_conv1_prep_weights +
    cova_conv1_fwd + cova_bn_finalize_fwd: same outputs and partial rows bit for bit, BatchNorm parameters to fp32 round-off."""
    from cova_web_object_detection_amd import engine
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 150, 330
    img = torch.rand(B, 3, H, W, generator=g).to(DEV)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(DEV)
    wk = torch.empty(154, 64, device=DEV)
    call("cova_conv1_prep_weights", w, wk)
    H1, W1 = query("cova_conv_out_size", H, 7, 2, 3), query("cova_conv_out_size", W, 7, 2, 3)
    n = query("cova_conv1_num_partials", B, H, W)
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(DEV), torch.randn(64, generator=g).to(DEV)
    cnt = float(B * H1 * W1)
    y, part = torch.empty(B, H1, W1, 64, device=DEV), torch.empty(n, 2, 64, device=DEV)
    call("cova_conv1_fwd", img, wk, y, part, B, H, W)
    ref = [torch.empty(64, device=DEV) for _ in range(4)]
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    call("cova_bn_finalize_fwd", part, n, 64, cnt, gamma, beta, rm, rv, None, 0.1, 1e-5, *ref)
    for _ in range(2):
        got = [torch.empty(64, device=DEV) for _ in range(4)]
        rm2, rv2 = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        tail = engine.BnTail(1, img, cnt, gamma=gamma, beta=beta, running_mean=rm2, running_var=rv2, num_batches_tracked=None,
                             scale=got[0], shift=got[1], mean=got[2], invstd=got[3])
        y2, part2 = torch.empty_like(y), torch.empty_like(part)
        call("cova_conv1_fwd_tail", img, w, y2, part2, B, H, W, tail.ptr)      # (the OIHW weight: no prep launch)
        assert torch.equal(y, y2) and torch.equal(part, part2)
        for a, b_ in zip(got + [rm2, rv2], ref + [rm, rv]):
            close(a, b_, 1e-6, "conv1 tail")


@pytest.mark.parametrize("geo", [1, 2])
@pytest.mark.parametrize("B,H,W,cap", [(1, 8, 32, 0), (2, 19, 45, 0), (3, 100, 200, 5)])
def test_conv3x3_winograd_inference_epilogue(B, H, W, cap, geo):
    """cova_conv3x3_wino_bnact == cova_conv3x3_wino followed by cova_bn_act_fwd (BatchNorm with given
    scale/shift, optional residual, optional ReLU), bit-exact, and against torch-CPU."""
    g = torch.Generator().manual_seed(11 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    xc, rc = rnd(B, 64, H, W), rnd(B, 64, H, W)
    x, res = nhwc(xc), nhwc(rc)
    w = rnd(64, 64, 3, 3) * 0.05
    sc, sh = rnd(64) * 0.5 + 1.0, rnd(64) * 0.3
    uf, ud = torch.empty(16, 16, 4, 64, device=DEV), torch.empty(16, 16, 4, 64, device=DEV)
    call("cova_conv3x3_prep_weights_wino", w.to(DEV), uf, ud)
    query("cova_set_option", 6, geo)
    query("cova_set_option", 2, cap)
    R = B * H * W
    try:
        conv = torch.empty_like(x)
        call("cova_conv3x3_wino", x, uf, None, None, None, None, None, conv, None, B, H, W)
        for addend, relu in ((None, 1), (res, 1), (None, 0), (res, 0)):
            ref, out = torch.empty_like(x), torch.empty_like(x)
            call("cova_bn_act_fwd", conv, 64, sc.to(DEV), sh.to(DEV), addend, 64 if addend is not None else 0,
                 ref, 64, R, 64, relu)
            call("cova_conv3x3_wino_bnact", x, uf, addend, sc.to(DEV), sh.to(DEV), relu, out, B, H, W)
            assert torch.equal(out, ref), (addend is not None, relu)
        t = F.conv2d(xc, w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + rc
        close(out.permute(0, 3, 1, 2), t, 2e-5, "inference epilogue vs torch")     # last case: +res, no relu
    finally:
        query("cova_set_option", 2, 0)
        query("cova_set_option", 6, 1)


@pytest.mark.parametrize("B,H,W", [(1, 8, 32), (2, 19, 45), (3, 100, 200)])
def test_conv3x3_winograd_f4x4_inference_epilogue(B, H, W):
    """cova_conv3x3_wino4_bnact (the eval-mode forward's launches, train.py:99-129) == the F(4x4,3x3) convolution (with
    or without the BatchNorm+ReLU-on-load prologue) followed by cova_bn_act_fwd (scale/shift, optional residual,
    optional ReLU), bit-exact; and against torch-CPU."""
    g = torch.Generator().manual_seed(13 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    xc, rc = rnd(B, 64, H, W), rnd(B, 64, H, W)
    x, res = nhwc(xc), nhwc(rc)
    w = rnd(64, 64, 3, 3) * 0.05
    sc, sh = (rnd(64) * 0.5 + 1.0).to(DEV), (rnd(64) * 0.3).to(DEV)
    abc = rnd(3, 64).to(DEV)
    uf, ud = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV)
    call("cova_conv3x3_wino4_prep", w.to(DEV), uf, ud)
    R = B * H * W
    for pro in (None, abc):
        conv = torch.empty_like(x)
        if pro is None:
            call("cova_conv3x3_wino4", x, uf, conv, None, B, H, W)
        else:
            call("cova_conv3x3_wino4_pro", x, pro, 1, uf, conv, None, B, H, W)
        for addend, relu in ((None, 1), (res, 1), (None, 0), (res, 0)):
            ref, out = torch.empty_like(x), torch.empty_like(x)
            call("cova_bn_act_fwd", conv, 64, sc, sh, addend, 64 if addend is not None else 0, ref, 64, R, 64, relu)
            call("cova_conv3x3_wino4_bnact", x, pro, 1, uf, addend, sc, sh, relu, out, B, H, W)
            assert torch.equal(out, ref), (pro is not None, addend is not None, relu)
    a1 = torch.relu(xc * abc[0].cpu().view(1, -1, 1, 1) + abc[2].cpu().view(1, -1, 1, 1))
    t = F.conv2d(a1.double(), w.double(), padding=1) * sc.cpu().double().view(1, -1, 1, 1) + sh.cpu().double().view(1, -1, 1, 1) + rc
    close(out.permute(0, 3, 1, 2), t, 2e-5, "F(4x4) inference epilogue vs torch (prologue, +res, no relu)")


@pytest.mark.parametrize("B,H,W,cap", [(1, 8, 32, 0), (1, 4, 4, 0), (2, 19, 45, 0), (3, 100, 200, 0), (3, 100, 200, 6),
                                       (2, 163, 37, 2)])
def test_conv3x3_winograd_f4x4_wgrad(B, H, W, cap):
    """F(4x4,3x3) weight gradient (csrc/conv_wgrad4.hip) against torch-CPU in fp64: sizes that are not multiples of the
    4 x 16-pixel K-step (ragged right / bottom tiles, a map smaller than one tile), few blocks walking many segments
    (grid cap) and the usual one; affine-on-load operands against the same launch on pre-transformed tensors; two
    launches bit-identical (fixed summation order)."""
    g = torch.Generator().manual_seed(5 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    xc, dzc = rnd(B, 64, H, W), rnd(B, 64, H, W)
    wr = (rnd(64, 64, 3, 3) * 0.05).double().requires_grad_(True)
    (F.conv2d(xc.double(), wr, padding=1) * dzc.double()).sum().backward()
    x, dy, z = nhwc(xc), nhwc(dzc), nhwc(rnd(B, 64, H, W))
    abc_a, abc_d = rnd(3, 64).to(DEV), rnd(3, 64).to(DEV)
    R = B * H * W
    query("cova_set_option", 2, cap)
    try:
        ws = torch.empty(query("cova_conv3x3_wgrad4_workspace_floats", B, H, W), device=DEV)
        nblk = query("cova_conv3x3_wgrad4_num_partials", B, H, W)
        assert ws.numel() == nblk * 9 * 4096                                   # transformed partial sums
        dw = torch.zeros(64, 64, 3, 3, device=DEV)
        call("cova_conv3x3_wgrad4", x, dy, dw, ws, B, H, W)
        close(dw, wr.grad, 5e-5, "F(4x4) wgrad vs fp64")      # (random data: no coherent signal; the F(2x2) test allows 2e-4)
        dw2 = torch.zeros_like(dw)
        call("cova_conv3x3_wgrad4", x, dy, dw2, ws, B, H, W)
        assert torch.equal(dw, dw2)
        # operands transformed on load == the plain launch on materialised operands (same fma, same order: bit-exact)
        a1 = torch.empty_like(x)
        call("cova_bn_act_fwd", x, 64, abc_a[0], abc_a[2], None, 0, a1, 64, R, 64, 1)
        dz2 = torch.addcmul(torch.addcmul(abc_d[2].expand_as(dy), z, abc_d[1]), dy, abc_d[0])     # A*dy + (B*z + C)
        dz1 = torch.addcmul(abc_d[2].expand_as(dy), dy, abc_d[0])                                 # A*dy + C
        ws2 = torch.empty_like(ws)
        for pa, pd in ((True, 2), (True, 0), (False, 2), (False, 1), (True, 1)):
            ref, got = torch.zeros_like(dw), torch.zeros_like(dw)
            call("cova_conv3x3_wgrad4", a1 if pa else x, {0: dy, 1: dz1, 2: dz2}[pd], ref, ws, B, H, W)
            dzo = torch.full_like(dy, float("nan")) if pd else None
            call("cova_conv3x3_wgrad4_partial", x, abc_a if pa else None, 1, dy, z if pd == 2 else None,
                 abc_d if pd else None, dzo, ws2, B, H, W)
            call("cova_conv3x3_wgrad4_finish", ws2, got, None, None, None, None, None, None, B, H, W)
            if pd:          # the side output = the operand the data-gradient kernel's prologue would form: same fma, every pixel once
                assert torch.isfinite(dzo).all(), "side output has unwritten pixels (pd = %d)" % pd
                close(dzo, dz2 if pd == 2 else dz1, 1e-6, "materialised gradient operand (pd = %d)" % pd)
            # (pd != 0: torch.addcmul rounds the products, the kernel's fma does not -- an ulp on the operand)
            close(got, ref, 2e-6 if pd == 0 else 1e-5, "F(4x4) wgrad affine-on-load %s %s" % (pa, pd))
        # one finish launch for two convolutions
        dwa, dwb = torch.zeros_like(dw), torch.zeros_like(dw)
        call("cova_conv3x3_wgrad4_partial", x, None, 0, dy, None, None, None, ws, B, H, W)
        call("cova_conv3x3_wgrad4_finish", ws, dwa, ws2, dwb, None, None, None, None, B, H, W)
        assert torch.equal(dwa, dw) and torch.equal(dwb, got)
    finally:
        query("cova_set_option", 2, 0)


@pytest.mark.parametrize("B,H,W,cap", [(1, 8, 32, 0), (2, 19, 45, 0), (3, 100, 200, 0), (3, 100, 200, 5)])
def test_conv3x3_winograd_wgrad_affine_on_load(B, H, W, cap):
    g = torch.Generator().manual_seed(3 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x, dy, z = nhwc(rnd(B, 64, H, W)), nhwc(rnd(B, 64, H, W)), nhwc(rnd(B, 64, H, W))
    abc_a, abc_d = rnd(3, 64).to(DEV), rnd(3, 64).to(DEV)
    R = B * H * W
    a1 = torch.empty_like(x)
    call("cova_bn_act_fwd", x, 64, abc_a[0], abc_a[2], None, 0, a1, 64, R, 64, 1)
    dz = torch.addcmul(torch.addcmul(abc_d[2].expand_as(dy), z, abc_d[1]), dy, abc_d[0])
    ws = torch.empty(query("cova_conv3x3_wgrad_workspace_floats", B, H, W), device=DEV)
    query("cova_set_option", 2, cap)
    try:
        ref, dw = torch.zeros(64, 64, 3, 3, device=DEV), torch.zeros(64, 64, 3, 3, device=DEV)
        for (pa, pd) in [(True, True), (True, False), (False, True), (False, False)]:
            call("cova_conv3x3_wgrad_wino", a1 if pa else x, dz if pd else dy, ref, ws, B, H, W)
            call("cova_conv3x3_wgrad_wino_pro", x, abc_a if pa else None, 1, dy, z if pd else None,
                 abc_d if pd else None, dw, ws, B, H, W)
            close(dw, ref, 2e-6, "wgrad affine-on-load %s %s" % (pa, pd))
        # two-step form: partial sums of two convolutions in two workspaces, ONE finish launch -- bit-identical
        ws2 = torch.empty_like(ws)
        dwa, dwb, refb = torch.zeros_like(dw), torch.zeros_like(dw), torch.zeros_like(dw)
        call("cova_conv3x3_wgrad_wino_pro", x, None, 0, dz, None, None, refb, ws, B, H, W)
        call("cova_conv3x3_wgrad_wino_partial", x, None, 0, dy, None, None, ws, B, H, W)
        call("cova_conv3x3_wgrad_wino_partial", x, None, 0, dz, None, None, ws2, B, H, W)
        call("cova_conv3x3_wgrad_wino_finish", ws, dwa, ws2, dwb, None, None, None, None, B, H, W)
        assert torch.equal(dwa, dw) and torch.equal(dwb, refb)
    finally:
        query("cova_set_option", 2, 0)


@pytest.mark.parametrize("B,H,W,cap", [(1, 32, 64, 0), (2, 70, 150, 0), (2, 150, 330, 0), (2, 150, 330, 7)])
def test_conv1_wgrad_with_pool_backward_folded_in(B, H, W, cap):
    """cova_conv1_wgrad_poolbwd == cova_bn_relu_maxpool_bwd_apply followed by cova_conv1_wgrad
    (same partial sums, same coefficients), incl. odd map sizes and windows across tile borders."""
    g = torch.Generator().manual_seed(H * 3 + W)
    x = torch.rand(B, 3, H, W, generator=g).to(DEV)
    H1, W1 = query("cova_conv_out_size", H, 7, 2, 3), query("cova_conv_out_size", W, 7, 2, 3)
    H2, W2 = query("cova_conv_out_size", H1, 3, 2, 1), query("cova_conv_out_size", W1, 3, 2, 1)
    y = nhwc(torch.randn(B, 64, H1, W1, generator=g))
    scale = (torch.rand(64, generator=g) - 0.3).to(DEV)
    shift = (torch.randn(64, generator=g) * 0.2).to(DEV)
    mean, invstd = (torch.randn(64, generator=g) * 0.1).to(DEV), (torch.rand(64, generator=g) + 0.5).to(DEV)
    p1, idx = torch.empty(B, H2, W2, 64, device=DEV), torch.empty(B, H2, W2, 64, device=DEV, dtype=torch.uint8)
    call("cova_bn_relu_maxpool_fwd", y, scale, shift, p1, idx, None, B, H1, W1)
    dp = nhwc(torch.randn(B, 64, H2, W2, generator=g)) * (p1 > 0)            # ReLU mask already applied
    npart = query("cova_bn_relu_maxpool_bwd_num_partials", B, H1, W1)
    part = torch.empty(npart, 2, 64, device=DEV)
    call("cova_bn_relu_maxpool_bwd_reduce", dp, idx, y, scale, shift, mean, invstd, part, B, H1, W1)
    coef, abc = torch.empty(2, 64, device=DEV), torch.empty(3, 64, device=DEV)
    call("cova_bn_finalize_bwd", part, npart, 64, float(B * H1 * W1), None, None, coef)
    call("cova_bn_finalize_bwd_abc", part, npart, 64, float(B * H1 * W1), None, None, mean, invstd, scale, abc)
    dy1 = torch.empty(B, H1, W1, 64, device=DEV)
    call("cova_bn_relu_maxpool_bwd_apply", dp, idx, y, scale, shift, mean, invstd, coef, dy1, B, H1, W1)
    ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=DEV)
    ref, dw = torch.zeros(64, 3, 7, 7, device=DEV), torch.zeros(64, 3, 7, 7, device=DEV)
    query("cova_set_option", 2, cap)
    try:
        call("cova_conv1_wgrad", x, dy1, ref, ws, B, H, W)
        call("cova_conv1_wgrad_poolbwd", x, y, dp, idx, abc, dw, ws, B, H, W)
    finally:
        query("cova_set_option", 2, 0)
    close(dw, ref, 2e-5, "conv1 wgrad with folded pool backward")


# ------------------------------------------------------------------------------------ deterministic backward
def _gat_case(N, K, Fd, D, ctx, seed):
    rs = np.random.RandomState(seed)
    h = torch.from_numpy(rs.standard_normal((N, Fd)).astype(np.float32))
    sd = {"gat.W_i.weight": torch.from_numpy(rs.uniform(-0.2, 0.2, (D, Fd)).astype(np.float32)),
          "gat.W_j.weight": torch.from_numpy(rs.uniform(-0.2, 0.2, (D, Fd)).astype(np.float32)),
          "gat.attention_layer.weight": torch.from_numpy(rs.uniform(-0.5, 0.5, (1, 2 * D)).astype(np.float32)),
          "gat.attention_layer.bias": torch.from_numpy(rs.uniform(-0.1, 0.1, 1).astype(np.float32))}
    g = torch.from_numpy(rs.standard_normal((N, D)).astype(np.float32))
    return h, sd, g


@pytest.mark.parametrize("kind", ["window", "random", "hub"])
def test_gat_backward_gather_is_deterministic_for_arbitrary_graphs(kind):
    """The transposed-index (gather) backward against autograd of the reference formulation, for the dataset's
    +-cs windows AND for index tables the reference's API equally accepts (models.py:171-177): random ids that
    cross pages and repeat inside a row, and a hub node named by every other node (in-degree >> 64)."""
    N, K, Fd, D = 157, 24, 40, 96
    rs = np.random.RandomState(7)
    if kind == "window":
        ctx = torch.from_numpy(synthetic.collate_context([synthetic.context_window_indices(n, 12) for n in (90, 11, 56)]))
    elif kind == "random":
        c = rs.randint(-1, N, (N, K))
        c[5] = -1
        c[9, :] = 3                                          # the same neighbour in every slot
        ctx = torch.from_numpy(c.astype(np.int64))
    else:
        c = rs.randint(0, N, (N, K))
        c[:, 0] = 42                                         # in-degree 157 + ... for node 42
        c[:, 5] = 42
        ctx = torch.from_numpy(c.astype(np.int64))
    h, sd, g = _gat_case(N, K, Fd, D, ctx, 3)
    hr = h.clone().requires_grad_(True)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (O.gat(hr, ctx, sdr) * g).sum().backward()
    params = {k: v.to(DEV) for k, v in sd.items()}
    outs = []
    for rep in range(2):
        hp = torch.empty(N, D, device=DEV)
        sv = engine.gat_fwd(h.to(DEV), Fd, N, Fd, ctx.to(DEV), params, hp, D)
        dh = torch.empty(N, Fd, device=DEV)
        grads = engine.gat_bwd(sv, g.to(DEV), D, params, dh, Fd, False)
        outs.append((dh, grads))
    close(outs[0][0], hr.grad, 1e-4, "gat dh (%s)" % kind)
    for k in sd:
        close(outs[0][1][k], sdr[k].grad.view_as(outs[0][1][k].cpu()), 1e-4, k)
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k        # bit-identical reruns
    assert torch.equal(outs[0][0], outs[1][0])
    # the transposed index itself: row j lists exactly the flat slots naming j, ascending
    csr = engine.gat_transpose(ctx.to(DEV)).cpu().numpy()
    flat = ctx.numpy().reshape(-1)
    for j in (0, 3, 42, N - 1):
        lo, hi = csr[j], csr[j + 1]
        assert np.array_equal(csr[N + 1 + lo:N + 1 + hi], np.nonzero(flat == j)[0])
    assert csr[N] == int((flat >= 0).sum())
    # the reused-workspace form (3 launches, counters left zero) equals the memset form, call after call
    E, used = N * K, N + 1 + int(csr[N])                    # (slots behind the last edge are never written)
    one = torch.empty(query("cova_gat_transpose_ints", N, K), dtype=torch.int32, device=DEV)
    call("cova_gat_transpose", ctx.to(DEV), N, K, one)
    assert np.array_equal(one.cpu().numpy()[:used], csr[:used])
    assert np.array_equal(engine.gat_transpose(ctx.to(DEV)).cpu().numpy()[:used], csr[:used])
    assert int(engine.gat_transpose(ctx.to(DEV))[N + 1 + 2 * E:].abs().sum()) == 0


@pytest.mark.parametrize("D,K", [(96, 24), (384, 24), (64, 1), (200, 64), (330, 48), (512, 7), (40, 13)])
def test_gat_wide_kernels_equal_the_chunk_kernels_bit_for_bit(D, K):
    """cova_set_option(16, .): the GAT forward / backward kernels with every 64-channel chunk of a neighbour row in flight
    (K <= 64, D <= 512) against the chunk-at-a-time kernels: the same sums in the same order."""
    N, Fd = 211, 72
    rs = np.random.RandomState(D + K)
    c = rs.randint(-1, N, (N, K))
    c[3] = -1
    c[:, 0] = 17                                             # a hub
    ctx = torch.from_numpy(c.astype(np.int64))
    h, sd, g = _gat_case(N, K, Fd, D, ctx, 5)
    params = {k: v.to(DEV) for k, v in sd.items()}
    res = []
    for wide in (0, 1):
        query("cova_set_option", 16, wide)
        try:
            hp = torch.empty(N, D, device=DEV)
            sv = engine.gat_fwd(h.to(DEV), Fd, N, Fd, ctx.to(DEV), params, hp, D)
            dh = torch.empty(N, Fd, device=DEV)
            grads = engine.gat_bwd(sv, g.to(DEV), D, params, dh, Fd, False)
            res.append((hp, sv["attn"], dh, grads))
        finally:
            query("cova_set_option", 16, 1)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for k in res[0][3]:
        assert torch.equal(res[0][3][k], res[1][3][k]), k
    assert bool(torch.isfinite(res[1][0]).all()) and bool(torch.isfinite(res[1][2]).all())


def test_roipool_backward_rows_are_deterministic_and_cover_the_map():
    """Heavily nested boxes (every box inside the previous one: maximal arg-max collisions), C = 256, a map
    wider than one LDS segment: equals the oracle's sequential scatter, no element left unwritten, reruns equal."""
    rs = np.random.RandomState(21)
    B, C, H, W = 2, 256, 9, 700
    feat = torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32))
    n = 40
    rois = np.zeros((n, 5), np.float32)
    for i in range(n):
        rois[i] = [i % B, 4.0 * i, 0.5 * i, 4 * W - 6.0 * i, 4 * H - 0.4 * i]
    rois = torch.from_numpy(rois)
    ref, ref_arg = O.roi_pool_argmax(feat, rois, (3, 3), 0.25)
    gout = torch.from_numpy(rs.standard_normal((n, C * 9)).astype(np.float32))
    gref = torch.empty(B, C, H, W)
    O._lib().oracle_roipool_bwd(O._fp(gout), O._fp(rois), O._fp(ref_arg), n, B, C, H, W, 3, 3, O._fp(gref))
    arg = ref_arg.reshape(n, C * 9).to(DEV)
    res = []
    for rep in range(2):
        gfeat = torch.full((B, H, W, C), float("nan"), device=DEV)
        call("cova_roipool_bwd", gout.to(DEV), C * 9, rois.to(DEV), arg, n, B, C, H, W, 3, 3, 0.25, gfeat, torch.empty(query("cova_roipool_bwd_workspace_words", n, B, C, 3, 3), dtype=torch.int32, device=DEV))
        res.append(gfeat)
    assert torch.isfinite(res[0]).all()
    close(nchw(res[0]), gref, 1e-5, "roipool bwd rows")
    assert torch.equal(res[0], res[1])
    assert int((gref != 0).sum()) < n * C * 9                     # collisions did happen


@pytest.mark.parametrize("ph,pw", [(1, 1), (2, 5), (7, 7)])
def test_roipool_other_output_sizes(ph, pw):
    """`-r` (roi_output_size, utils.py:20) other than the default 3: forward bit-exact vs the C oracle, backward
    (generic bin loop of the row-owner kernel) vs its sequential scatter, masked variant consistent."""
    rs = np.random.RandomState(ph * 10 + pw)
    B, C, H, W = 2, 64, 30, 45
    feat = torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32))
    n = 60
    x1 = rs.uniform(-10, 150, n); y1 = rs.uniform(-10, 100, n)
    rois = torch.from_numpy(np.stack([rs.randint(0, B, n), x1, y1, x1 + rs.uniform(1, 120, n), y1 + rs.uniform(1, 90, n)], 1)
                            .astype(np.float32))
    nb = ph * pw
    ref, ref_arg = O.roi_pool_argmax(feat, rois, (ph, pw), 0.25)
    out = torch.empty(n, C * nb, device=DEV)
    arg = torch.empty((n, C * nb), device=DEV, dtype=torch.int32)
    call("cova_roipool_fwd", nhwc(feat), rois.to(DEV), n, B, C, H, W, ph, pw, 0.25, out, C * nb, arg)
    assert torch.equal(out.cpu(), ref.reshape(n, -1)) and torch.equal(arg.cpu(), ref_arg.reshape(n, -1))
    gout = torch.from_numpy(rs.standard_normal((n, C * nb)).astype(np.float32))
    gref = torch.empty(B, C, H, W)
    O._lib().oracle_roipool_bwd(O._fp(gout), O._fp(rois), O._fp(ref_arg), n, B, C, H, W, ph, pw, O._fp(gref))
    ws = torch.empty(query("cova_roipool_bwd_workspace_words", n, B, C, ph, pw), dtype=torch.int32, device=DEV)
    g = torch.full((B, H, W, C), float("nan"), device=DEV)
    call("cova_roipool_bwd", gout.to(DEV), C * nb, rois.to(DEV), arg, n, B, C, H, W, ph, pw, 0.25, g, ws)
    close(nchw(g), gref, 1e-5, "roipool bwd %dx%d" % (ph, pw))
    # whole model with that output size (eval logits vs the oracle)
    cfg = dict(roi_output_size=(ph, pw), n_classes=4, use_context=True, hidden_dim=32, bbox_hidden_dim=8,
               n_additional_feat=0, drop_prob=0.0)
    from cova_web_object_detection_amd import weights
    sd = weights.seeded_state_dict(3, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    batch = synthetic.make_batch(2, img_h=64, boxes_per_page=[9, 12], context_size=4, seed=3)
    params = {k: v.to(DEV) for k, v in sd.items() if k in O.param_keys(sd)}
    buffers = {k: v.to(DEV) for k, v in sd.items() if k not in params}
    args = [batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")]
    logits, sv = engine.model_fwd(cfg, params, buffers, *args, True)
    loss, dl, _ = engine.ce_sum(logits, batch["labels"].to(DEV))
    grads = engine.model_bwd(sv, dl, params)
    ref_l = O.forward(O.clone_state_dict(sd), batch["images"], batch["bboxes"], batch["additional_feats"],
                      batch["context_indices"], cfg, True)
    close(logits, ref_l, 2e-4, "logits with %dx%d bins" % (ph, pw))
    assert all(torch.isfinite(v).all() for v in grads.values())


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(1, 16, 32), (2, 19, 45), (1, 40, 70), (1, 3, 5), (3, 100, 200)])
def test_conv3x3_winograd_f4x4(B, H, W):
    """F(4x4,3x3) form (csrc/conv_wino4.hip): forward with statistics and the data-gradient weights, against
    torch's conv2d on CPU.  Its fp32 error is ~10x F(2x2,3x3)'s (transform constants up to 8 and 1/24), still
    two orders inside the 1e-4 gate."""
    g = torch.Generator().manual_seed(H * W + 1)
    x = torch.randn(B, 64, H, W, generator=g).clamp_min(0)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    uf, ud = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV)
    call("cova_conv3x3_wino4_prep", w.to(DEV), uf, ud)
    if B == 3:
        query("cova_set_option", 2, 5)          # several tiles per persistent block
    n = query("cova_conv3x3_wino4_num_partials", B, H, W)
    out, part = torch.full((B, H, W, 64), 7.0, device=DEV), torch.empty(n, 2, 64, device=DEV)
    call("cova_conv3x3_wino4", nhwc(x), uf, out, part, B, H, W)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    close(nchw(out), ref, 2e-5, "wino4 fwd")
    close(part[:, 0].double().sum(0), ref.sum((0, 2, 3)), 1e-4, "wino4 stat sum")
    close(part[:, 1].double().sum(0), (ref * ref).sum((0, 2, 3)), 1e-4, "wino4 stat sumsq")
    dz = torch.randn(B, 64, H, W, generator=g)
    xr = x.double().clone().requires_grad_(True)
    (F.conv2d(xr, w.double(), padding=1) * dz.double()).sum().backward()
    dx = torch.empty(B, H, W, 64, device=DEV)
    call("cova_conv3x3_wino4", nhwc(dz), ud, dx, None, B, H, W)
    close(nchw(dx), xr.grad, 2e-5, "wino4 dgrad")
    # affine + ReLU on load (zero padding must stay zero although relu(C) != 0)
    z = torch.randn(B, 64, H, W, generator=g)
    abc = torch.stack([torch.rand(64, generator=g) + 0.5, torch.zeros(64), torch.randn(64, generator=g) * 0.5 + 0.3])
    for relu in (1, 0):
        ain = abc[0].view(1, 64, 1, 1) * z + abc[2].view(1, 64, 1, 1)
        if relu:
            ain = ain.clamp_min(0)
        refp = F.conv2d(ain.double(), w.double(), padding=1)
        o3, p3 = torch.empty(B, H, W, 64, device=DEV), torch.empty(n, 2, 64, device=DEV)
        call("cova_conv3x3_wino4_pro", nhwc(z), abc.to(DEV), relu, uf, o3, p3, B, H, W)
        close(nchw(o3), refp, 2e-5, "wino4 prologue relu=%d" % relu)
        close(p3[:, 0].double().sum(0), refp.sum((0, 2, 3)), 1e-4, "wino4 prologue stat sum")
    query("cova_set_option", 2, 0)


# ------------------------------------------------------------------------------------ conv1 on the bf16 matrix pipe
@pytest.mark.parametrize("B,H,W,cap", [(2, 250, 333, 0), (3, 150, 330, 5)])
def test_conv1_bf16_split_error_class(B, H, W, cap):
    """conv1 forward and weight gradient as six bf16-MFMA products of three-piece operands (csrc/conv.hip,
    conv1_7x7_bf3_kernel / conv1_wgrad_bf3_kernel) against a float64 convolution: the error is the f32-MFMA kernels'
    (cova_set_option(7, 1)), not a reduced-precision one; results are bit-reproducible; edge and multi-tile paths."""
    g = torch.Generator().manual_seed(H + 7 * W)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    wr = w.double().requires_grad_(True)
    ref = F.conv2d(x.double(), wr, stride=2, padding=3)
    H1, W1 = ref.shape[2], ref.shape[3]
    dy = torch.randn(B, 64, H1, W1, generator=g)
    (ref * dy.double()).sum().backward()
    xg, wg, dyg = x.to(DEV), w.to(DEV), nhwc(dy)
    ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=DEV)
    err = {}
    query("cova_set_option", 2, cap)
    try:
        for f32 in (1, 0):
            query("cova_set_option", 7, f32)
            outs = []
            for _ in range(2):
                nt = query("cova_conv1_num_partials", B, H, W)
                out, part = torch.zeros(B, H1, W1, 64, device=DEV), torch.zeros(nt, 2, 64, device=DEV)
                call("cova_conv1_fwd_tail", xg, wg, out, part, B, H, W, None)
                dw = torch.zeros(64, 3, 7, 7, device=DEV)
                call("cova_conv1_wgrad", xg, dyg, dw, ws, B, H, W)
                outs.append((out, part, dw))
            assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "not bit-reproducible"
            out, part, dw = outs[0]
            e_y = float((nchw(out).double().cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
            e_w = float((dw.double().cpu() - wr.grad).abs().max() / wr.grad.abs().max())
            e_q = float((part[:, 1].double().sum(0).cpu() - (ref.detach() ** 2).sum((0, 2, 3))).abs().max()
                        / (ref.detach() ** 2).sum((0, 2, 3)).max())
            err[f32] = (e_y, e_w, e_q)
    finally:
        query("cova_set_option", 7, 0)
        query("cova_set_option", 2, 0)
    print("conv1 error against fp64 (output, weight gradient, channel sums of squares): f32 MFMA %.2e %.2e %.2e | "
          "bf16 split %.2e %.2e %.2e" % (err[1] + err[0]))
    for a, b in zip(err[0], err[1]):
        assert a <= 2.0 * b + 2e-7, (err[0], err[1])
    assert err[0][0] < 2e-6 and err[0][1] < 2e-6


@pytest.mark.parametrize("B,H,W,cap,relu_in", [(1, 8, 32, 0, False), (2, 19, 45, 0, True), (3, 100, 200, 5, True), (2, 320, 320, 0, True)])
def test_conv3x3_winograd_f4x4_split_error_class(B, H, W, cap, relu_in):
    """The F(4x4,3x3) forward / data-gradient launches with the transform-domain products as six bf16-MFMA products of
    three-piece operands (csrc/conv_wino4_split.h, the default) against a float64 convolution: the error is the f32-MFMA main
    loop's (cova_set_option(9, 1)), not a reduced-precision one; the statistics rows agree; the ReLU masks of the data-gradient
    epilogue are identical; results are bit-reproducible; one-tile, ragged, multi-tile-per-block and full-width cases."""
    g = torch.Generator().manual_seed(3 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    xc = rnd(B, 64, H, W)
    if relu_in:
        xc = xc.clamp_min(0)
    x, add, z = nhwc(xc), nhwc(rnd(B, 64, H, W)), nhwc(rnd(B, 64, H, W))
    w = rnd(64, 64, 3, 3) * 0.05
    abc = rnd(3, 64).to(DEV)
    msc, msh = rnd(64).to(DEV), rnd(64).to(DEV) * 0.3
    mean, invstd = rnd(64).to(DEV) * 0.2, (torch.rand(64, generator=g) + 0.5).to(DEV)
    nu = query("cova_conv3x3_wino4_u_floats")
    uf, ud = torch.empty(nu, device=DEV), torch.empty(nu, device=DEV)
    call("cova_conv3x3_wino4_prep", w.to(DEV), uf, ud)
    ref_f = F.conv2d(xc.double(), w.double(), padding=1)
    ref_d = F.conv2d(xc.double(), w.double().flip(2, 3).transpose(0, 1), padding=1)
    pre = torch.relu(abc[0].cpu().double().view(1, -1, 1, 1) * xc.double() + abc[2].cpu().double().view(1, -1, 1, 1))
    ref_p = F.conv2d(pre, w.double(), padding=1)
    N = None
    err, keep = {}, {}
    query("cova_set_option", 2, cap)
    try:
        n4 = query("cova_conv3x3_wino4_num_partials", B, H, W)
        for f32 in (1, 0):
            query("cova_set_option", 9, f32)
            runs = []
            for _ in range(2):
                o_f, p_f = torch.zeros_like(x), torch.zeros(n4, 2, 64, device=DEV)
                call("cova_conv3x3_wino4_full", x, N, N, 0, uf, N, N, N, N, N, N, N, o_f, p_f, B, H, W)
                o_p, p_p = torch.zeros_like(x), torch.zeros(n4, 2, 64, device=DEV)
                call("cova_conv3x3_wino4_full", x, N, abc, 1, uf, N, N, N, N, N, N, N, o_p, p_p, B, H, W)
                o_d, p_d = torch.zeros_like(x), torch.zeros(n4, 2, 64, device=DEV)
                call("cova_conv3x3_wino4_full", x, N, N, 0, ud, add, N, msc, msh, z, mean, invstd, o_d, p_d, B, H, W)
                runs.append((o_f, p_f, o_p, p_p, o_d, p_d))
            assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1])), "not bit-reproducible"
            o_f, p_f, o_p, p_p, o_d, p_d = runs[0]
            rel = lambda got, ref: float((nchw(got).double().cpu() - ref).abs().max() / ref.abs().max())
            e_q = float((p_f[:, 1].double().sum(0).cpu() - (ref_f ** 2).sum((0, 2, 3))).abs().max() / (ref_f ** 2).sum((0, 2, 3)).max())
            gate = (nchw(z).cpu().double() * msc.cpu().double().view(1, -1, 1, 1) + msh.cpu().double().view(1, -1, 1, 1)) > 0
            e_d = float(((nchw(o_d).double().cpu() - (ref_d + nchw(add).double().cpu()) * gate).abs() * gate).max() / ref_d.abs().max())
            err[f32] = (rel(o_f, ref_f), rel(o_p, ref_p), e_q, e_d)
            keep[f32] = o_d
            if not f32 and B * ((H + 7) // 8) * ((W + 31) // 32) >= 200:
                # the bf16 MFMA's accumulation is not sign-symmetric (every output 7e-8 of the mean magnitude low when all tiles
                # multiply with +U); tiles of alternating sign make the offset cancel in sums over the map: the mean SIGNED
                # error must be at the f32 loop's level (measured 1e-9), not 7e-8
                bias = float((nchw(o_f).double().cpu() - ref_f).sum() / ref_f.abs().sum())
                assert abs(bias) < 1.5e-8, "coherent offset of the split loop's outputs: %.2e of the mean magnitude" % bias
    finally:
        query("cova_set_option", 9, 0)
        query("cova_set_option", 2, 0)
    print("F(4x4,3x3) error against fp64 (forward, BatchNorm+ReLU on load, channel sums of squares, data gradient): f32 MFMA "
          "%.2e %.2e %.2e %.2e | bf16 split %.2e %.2e %.2e %.2e" % (err[1] + err[0]))
    assert torch.equal(keep[0] == 0, keep[1] == 0), "ReLU masks of the two main loops differ"
    for a, b in zip(err[0], err[1]):
        assert a <= 1.5 * b + 2e-7, (err[0], err[1])
    assert max(err[0][0], err[0][1], err[0][3]) < 2e-5


def test_conv1_kernels_do_not_depend_on_the_batch_partition():
    """The conv1 kernels on a 4-page batch in one call against two 2-page calls with bit-identical operands: forward
    outputs and statistics rows are bitwise the same, the weight gradient (BatchNorm + ReLU + MaxPool backward folded
    in) is the sum of the halves' to fp32 re-association -- what the data-parallel / SyncBN tests rely on (their own
    residual differences are decisions that fall the other way, tests/test_syncbn_gpu.py)."""
    g = torch.Generator().manual_seed(11)
    B, H, W = 4, 128, 160
    x = torch.rand(B, 3, H, W, generator=g).to(DEV)
    H1, W1 = query("cova_conv_out_size", H, 7, 2, 3), query("cova_conv_out_size", W, 7, 2, 3)
    H2, W2 = query("cova_conv_out_size", H1, 3, 2, 1), query("cova_conv_out_size", W1, 3, 2, 1)
    y = (torch.randn(B, H1, W1, 64, generator=g) * 3).to(DEV)
    scale, shift = (torch.rand(64, generator=g) - 0.3).to(DEV), (torch.randn(64, generator=g) * 0.2).to(DEV)
    p1 = torch.empty(B, H2, W2, 64, device=DEV)
    idx = torch.empty(B, H2, W2, 64, device=DEV, dtype=torch.uint8)
    call("cova_bn_relu_maxpool_fwd", y, scale, shift, p1, idx, None, B, H1, W1)
    dp = torch.randn(B, H2, W2, 64, generator=g).to(DEV) * (p1 > 0)
    abc = (torch.randn(3, 64, generator=g) * 0.3).to(DEV)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).to(DEV)

    def wgrad(sl):
        n = sl.stop - sl.start
        ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", n, H, W), device=DEV)
        dw = torch.zeros(64, 3, 7, 7, device=DEV)
        call("cova_conv1_wgrad_poolbwd", x[sl].contiguous(), y[sl].contiguous(), dp[sl].contiguous(),
             idx[sl].contiguous(), abc, dw, ws, n, H, W)
        return dw

    def fwd(sl):
        n = sl.stop - sl.start
        out = torch.zeros(n, H1, W1, 64, device=DEV)
        part = torch.zeros(query("cova_conv1_num_partials", n, H, W), 2, 64, device=DEV)
        call("cova_conv1_fwd_tail", x[sl].contiguous(), w, out, part, n, H, W, None)
        return out, part

    full, halves = wgrad(slice(0, 4)), wgrad(slice(0, 2)) + wgrad(slice(2, 4))
    assert float((full - halves).abs().max()) <= 3e-7 * float(full.abs().max())
    of, pf = fwd(slice(0, 4))
    (o0, p0), (o1, p1_) = fwd(slice(0, 2)), fwd(slice(2, 4))
    assert torch.equal(of, torch.cat((o0, o1)))
    assert torch.equal(pf, torch.cat((p0, p1_)))          # (one tile per block at this size: the rows are per tile)


@pytest.mark.parametrize("B,H,W,cap", [(2, 150, 330, 0), (2, 150, 330, 7), (1, 70, 90, 0)])
def test_conv1_wgrad_role_split_equals_f32_form(B, H, W, cap):
    """The bf16-split conv1 weight gradient (role-split waves) and the f32-MFMA kernel (cova_set_option(7, 1)) on the same
    operands, pool backward folded in: equal to fp32 re-association, on maps with edge tiles, odd sizes and many tiles per
    block."""
    g = torch.Generator().manual_seed(H + W + cap)
    x = torch.rand(B, 3, H, W, generator=g).to(DEV)
    H1, W1 = query("cova_conv_out_size", H, 7, 2, 3), query("cova_conv_out_size", W, 7, 2, 3)
    H2, W2 = query("cova_conv_out_size", H1, 3, 2, 1), query("cova_conv_out_size", W1, 3, 2, 1)
    y = (torch.randn(B, H1, W1, 64, generator=g) * 2).to(DEV)
    scale, shift = (torch.rand(64, generator=g) - 0.3).to(DEV), (torch.randn(64, generator=g) * 0.2).to(DEV)
    p1 = torch.empty(B, H2, W2, 64, device=DEV)
    idx = torch.empty(B, H2, W2, 64, device=DEV, dtype=torch.uint8)
    call("cova_bn_relu_maxpool_fwd", y, scale, shift, p1, idx, None, B, H1, W1)
    dp = torch.randn(B, H2, W2, 64, generator=g).to(DEV) * (p1 > 0)
    abc = (torch.randn(3, 64, generator=g) * 0.3).to(DEV)
    ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=DEV)
    out = {}
    query("cova_set_option", 2, cap)
    try:
        for name, o7 in (("f32", 1), ("roles", 0)):
            query("cova_set_option", 7, o7)
            dw = torch.zeros(64, 3, 7, 7, device=DEV)
            call("cova_conv1_wgrad_poolbwd", x, y, dp, idx, abc, dw, ws, B, H, W)
            out[name] = dw
    finally:
        query("cova_set_option", 7, 0)
        query("cova_set_option", 2, 0)
    s = float(out["f32"].abs().max())
    assert float((out["roles"] - out["f32"]).abs().max()) <= 5e-7 * s


@pytest.mark.parametrize("B,H,W", [(1, 14, 18), (2, 37, 52), (1, 64, 64)])
def test_conv1_data_gradient_and_pool_backward_operand(B, H, W):
    """cova_conv1_dgrad against torch's autograd of conv2d(3, 64, 7, stride 2, pad 3) w.r.t. its input (fp64), odd sizes
    included; cova_pool_bwd_dy1 against the formula it restates, abc[0]*route(dp, idx) + abc[1]*y1 + abc[2], with the
    arg-max codes of cova_bn_relu_maxpool_fwd."""
    g = torch.Generator().manual_seed(7 * H + W)
    x = torch.randn(B, 3, H, W, generator=g).double().requires_grad_(True)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    y = F.conv2d(x, w.double(), stride=2, padding=3)
    H1, W1 = y.shape[2], y.shape[3]
    dy = torch.randn(B, 64, H1, W1, generator=g)
    (y * dy.double()).sum().backward()
    dimg = torch.full((B, 3, H, W), 7.0, device=DEV)
    call("cova_conv1_dgrad", nhwc(dy), w.to(DEV), dimg, B, H, W)
    close(dimg, x.grad, 2e-6, "conv1 data gradient")
    # ---- the materialised pool-backward operand
    H2, W2 = query("cova_conv_out_size", H1, 3, 2, 1), query("cova_conv_out_size", W1, 3, 2, 1)
    y1 = nhwc(torch.randn(B, 64, H1, W1, generator=g) * 2)
    scale, shift = (torch.rand(64, generator=g) - 0.3).to(DEV), (torch.randn(64, generator=g) * 0.2).to(DEV)
    p1 = torch.empty(B, H2, W2, 64, device=DEV)
    idx = torch.empty(B, H2, W2, 64, device=DEV, dtype=torch.uint8)
    call("cova_bn_relu_maxpool_fwd", y1, scale, shift, p1, idx, None, B, H1, W1)
    dp = nhwc(torch.randn(B, 64, H2, W2, generator=g)) * (p1 > 0)
    abc = (torch.randn(3, 64, generator=g) * 0.3).to(DEV)
    got = torch.full_like(y1, float("nan"))
    call("cova_pool_bwd_dy1", dp, idx, y1, abc, got, B, H1, W1)
    # reference: scatter every window's gradient to its arg-max position (code = ky*3 + kx, window rows 2*oy-1+ky)
    route = torch.zeros(B, H1 + 2, W1 + 2, 64, dtype=torch.float64)
    code = idx.cpu().long()
    ky, kx = code // 3, code % 3
    bb, oy, ox, cc = torch.meshgrid(torch.arange(B), torch.arange(H2), torch.arange(W2), torch.arange(64), indexing="ij")
    route.index_put_((bb, 2 * oy + ky, 2 * ox + kx, cc), dp.cpu().double(), accumulate=True)      # (+1 pad offset folded: 2*oy-1+ky+1)
    route = route[:, 1:H1 + 1, 1:W1 + 1]
    want = abc[0].cpu().double() * route + abc[1].cpu().double() * y1.cpu().double() + abc[2].cpu().double()
    assert torch.isfinite(got).all()
    close(got, want, 1e-6, "pool-backward operand")


def test_option_state_is_fixed_after_the_first_query():
    """cova_set_option: mutable until the library's first query or launch, refused afterwards (10001) unless the process opted in
    with COVA_ALLOW_OPTION_CHANGES=1 (this suite and bench.py do); setting a key to the value it has always succeeds."""
    import os
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('COVA_ALLOW_OPTION_CHANGES', None)\n"
            "import cova_amd; from cova_web_object_detection_amd import _lib; q = _lib.query\n"
            "r = [q('cova_set_option', 9, 1), q('cova_set_option', 9, 0), q('cova_set_option', 3, 1)]\n"
            "n = q('cova_conv3x3_wino4_num_partials', 2, 64, 64)\n"
            "r += [q('cova_set_option', 9, 0), q('cova_set_option', 9, 1), q('cova_set_option', 2, 5), q('cova_set_option', 2, 0)]\n"
            "print(r)" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "[0, 0, 10001, 0, 10001, 10001, 0]", out.stdout
