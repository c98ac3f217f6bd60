"""Debug: chunk vs wide GAT backward (source side) against an fp64 reference, and each against itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib, engine
call, query = _lib.call, _lib.query
DEV = "cuda:0"
for D, K in ((64, 64), (64, 24), (64, 32), (64, 36), (64, 48)):
    N = 211
    rs = np.random.RandomState(D + K)
    c = rs.randint(-1, N, (N, K)); c[3] = -1; c[:, 0] = 17
    ctx = torch.from_numpy(c.astype(np.int64)).to(DEV)
    Wh = torch.randn(N, 2 * D, device=DEV); g = torch.randn(N, D, device=DEV)
    aw = torch.randn(1, 2 * D, device=DEV) * 0.3; ab = torch.zeros(1, device=DEV)
    s, t, attn, hp = torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, K, device=DEV), torch.empty(N, D, device=DEV)
    call("cova_gat_fwd", Wh, 2 * D, aw, ab, ctx, N, K, D, 0.2, s, t, attn, hp, D)
    csr = engine.gat_transpose(ctx).clone()
    # fp64 reference of du
    jj = ctx.clamp(min=0)
    dal = torch.einsum("nd,nkd->nk", g.double(), Wh[:, D:].double()[jj])
    valid = ctx >= 0
    dal = dal * valid
    dot = (attn.double() * dal).sum(1, keepdim=True)
    u = s.double()[:, None] + t.double()[jj]
    du_ref = (attn.double() * (dal - dot) * torch.where(u > 0, 1.0, 0.2)) * valid
    res = []
    for wide in (0, 0, 1, 1):
        query("cova_set_option", 16, wide)
        dWh = torch.full((N, 2 * D), 7.0, device=DEV); ds, dt = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        daw, dab, du = torch.empty(1, 2 * D, device=DEV), torch.empty(1, device=DEV), torch.empty(N, K, device=DEV)
        call("cova_gat_bwd", g, D, Wh, 2 * D, s, t, attn, ctx, aw, N, K, D, 0.2, dWh, 2 * D, ds, dt, daw, dab, csr, du)
        res.append(du.clone())
    query("cova_set_option", 16, 1)
    sc = float(du_ref.abs().max())
    bad = (res[0] != res[2]).nonzero()
    print("D %d K %d: chunk==chunk %s  wide==wide %s  chunk==wide %s | err chunk %.2e wide %.2e | slots that differ: %s" % (
        D, K, torch.equal(res[0], res[1]), torch.equal(res[2], res[3]), torch.equal(res[0], res[2]),
        float((res[0].double() - du_ref).abs().max()) / sc, float((res[2].double() - du_ref).abs().max()) / sc,
        sorted(set(bad[:, 1].tolist()))[:70]))
