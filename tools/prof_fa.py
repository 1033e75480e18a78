"""Factor-analysis kernels at full CelebA size (N = 202 599, Kc = 50, views d = 20 / 40): HIP-event time per launch and the
algorithmic HBM bytes it moves, against the 8 TB/s peak; then whole FA-only iterations through the drop-in API (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fa-vae_amd"), os.path.join(ROOT, "fa-vae_amd", "condition_a_VAE")):
    sys.path.insert(0, p)
import numpy as np, torch
from favae_hip import fa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 202599
K, d0, d1 = 50, 20, 40
F64 = torch.float64
g = torch.Generator(device="cuda").manual_seed(0)
X0, X1, Z = (torch.randn(N, k, dtype=F64, device="cuda", generator=g) for k in (d0, d1, K))
sc = fa.Scratch("cuda")


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


rows = []
def rec(name, t, nbytes, flops):
    rows.append((name, t, nbytes, flops))
    print("%-46s %8.1f us  %7.1f MB  %6.0f GB/s (%4.1f %% of 8 TB/s)  %6.2f TFLOP/s f64" % (name, t * 1e6, nbytes / 1e6, nbytes / t / 1e9, 100 * nbytes / t / 8e12, flops / t / 1e12))

# (a matrix with itself: the rows are loaded once for both operands since round 3, and 10 of the 16 tiles are computed)
rec("atb  Z^T [Z | 1]            (K x K+1)", timeit(lambda: fa.atb(Z, Z, True, sc)), 8.0 * N * K, 2.0 * N * K * (K + 1))
rec("atb  X1^T [Z | 1]           (40 x K+1)", timeit(lambda: fa.atb(X1, Z, True, sc)), 8.0 * N * (d1 + K), 2.0 * N * d1 * (K + 1))
rec("atb  X0^T [X0 | 1]          (20 x 21)", timeit(lambda: fa.atb(X0, X0, True, sc)), 8.0 * N * d0, 2.0 * N * d0 * (d0 + 1))
P = torch.randn(d1, K, dtype=F64, device="cuda", generator=g); c = torch.randn(K, dtype=F64, device="cuda", generator=g)
out = torch.empty(N, K, dtype=F64, device="cuda")
rec("rows_gemm  out = a (X1 P + c) + b Z  (N x K)", timeit(lambda: fa.rows_gemm(X1, P, c, 0.7, Z, 0.3, out)), 8.0 * N * (d1 + 2 * K), 2.0 * N * d1 * K)
Wt = torch.randn(K, d1, dtype=F64, device="cuda", generator=g); b1 = torch.randn(d1, dtype=F64, device="cuda", generator=g)
zwb = torch.empty(N, d1, dtype=F64, device="cuda")
rec("rows_gemm  Z W^T + b         (N x 40)", timeit(lambda: fa.rows_gemm(Z, Wt, b1, 1.0, None, 0.0, zwb)), 8.0 * N * (K + d1), 2.0 * N * d1 * K)
t = (torch.rand(N, d1, device="cuda", generator=g) > 0.5).to(F64)
xi = torch.rand(N, d1, dtype=F64, device="cuda", generator=g) + 0.5
mean, cov = torch.empty_like(xi), torch.empty_like(xi)
rec("mult_update (update_x + update_xi fused)", timeit(lambda: fa.mult_update(t, zwb, xi.clone(), mean, cov, 1.3, sc)) - timeit(lambda: xi.clone()),
    8.0 * N * d1 * 6, 0.0)

# the batched SPD inverse (favae_chol_inv_batched): one launch for the inverses of an update_Z / update_w
for Kc, nb in ((50, 3), (50, 5), (100, 5), (200, 5), (200, 9)):
    G = torch.randn(nb, 2 * Kc, Kc, dtype=F64, device="cuda", generator=g)
    mats = [dict(terms=[(1.0, (G[i].t() @ G[i]).contiguous())], diag_add=1.0) for i in range(nb)]
    tt = timeit(lambda: fa.chol_inv_batched(mats, Kc, sc), reps=20)
    print("chol_inv_batched  %d x (%d x %d)  one launch %38.1f us   %6.2f GFLOP/s f64 per matrix (K^3 flop each)" % (nb, Kc, Kc, tt * 1e6, Kc ** 3 / tt / 1e9))

# whole FA-only iterations through the drop-in API ('reg' d = 20 + 40 labels), as SURVEY 6 timed the reference (2.8 s / update on 8 cores)
import favae
rs = np.random.RandomState(1)
zt = rs.randn(N, 5)
x0 = zt @ rs.randn(5, d0) + 0.1 * rs.randn(N, d0)
lab = (zt @ rs.randn(5, d1) > 0).astype(int)
np.random.seed(0); torch.manual_seed(0)
m = favae.SSHIBA(K, 1)
m.fit(m.struct_data(x0, "reg"), m.struct_data(lab, "mult"), max_iter=0, pruning_crit=1e-3, verbose=0, store=False, pretrained=1)
def it():
    m.update(iter_count=1); m.depruning(1e-15); m.L.append(m.update_bound()); m.pruning(1e-3)
it(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.time(); it(); torch.cuda.synchronize(); ts.append(time.time() - t0)
print("FA-only iteration (update + bound + pruning), N = %d, Kc = %d: %.1f ms (median of 5) = %.2f M samples/s; bound %.6e"
      % (N, K, 1e3 * float(np.median(ts)), N / float(np.median(ts)) / 1e6, m.L[-1]))
