"""cova_sgemm at the shapes of the hot path (N = 1440 boxes)."""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
call = _lib.call
dev = "cuda:0"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


N = 1440
shapes = [  # (tA, tB, M, N, K, what)
    (0, 1, N, 768, 608, "gat fwd   Wh = h [W_i;W_j]^T"),
    (0, 0, N, 608, 768, "gat bwd   dh = dWh [W_i;W_j]"),
    (1, 0, 768, 608, N, "gat bwd   dW = dWh^T h"),
    (0, 1, N, 992, 992, "decoder   y = x W^T"),
    (0, 0, N, 992, 992, "decoder   dx = dy W"),
    (1, 0, 992, 992, N, "decoder   dW = dy^T x"),
]
for tA, tB, M, Nn, K, what in shapes:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((Nn, K) if tB else (K, Nn), device=dev)
    C = torch.empty(M, Nn, device=dev)
    lda, ldb = A.shape[1], B.shape[1]
    _lib.query("cova_set_option", 22, 0)
    t0 = timeit(lambda: call("cova_sgemm", tA, tB, M, Nn, K, A, lda, B, ldb, C, Nn, None, 0))
    _lib.query("cova_set_option", 22, 1)
    t = timeit(lambda: call("cova_sgemm", tA, tB, M, Nn, K, A, lda, B, ldb, C, Nn, None, 0))
    ref = (A.t() if tA else A) @ (B.t() if tB else B)
    err = float((C - ref).abs().max() / ref.abs().max())
    t2 = timeit(lambda: torch.matmul(A.t() if tA else A, B.t() if tB else B))
    print("%-32s M%5d N%5d K%5d  %6.1f us  %5.1f TF/s  (register-staged kernel %6.1f us, torch/hipBLASLt %6.1f us)  rel err %.1e"
          % (what, M, Nn, K, t * 1e3, 2 * M * Nn * K / t / 1e9, t0 * 1e3, t2 * 1e3, err))
