"""Where the time of one whitening learning goes (whiten.py:37-53 at D = 2048 on 20 000 pairs of 40 000 descriptors): the
device GEMMs against the host's dense factorisations (Cholesky, inverse, eig -- on the host in the reference and here)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mdir_amd import whiten

D, n_pairs, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048), 20000, 40000
rng = np.random.default_rng(0)
X = rng.standard_normal((D, n)) * np.linspace(1.0, 0.05, D)[:, None] + rng.standard_normal((D, 1))
X /= np.linalg.norm(X, axis=0, keepdims=True)
q, p = rng.integers(0, n, n_pairs), rng.integers(0, n, n_pairs)


def lap(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print("%-46s %9.1f ms" % (name, (time.perf_counter() - t) * 1e3), flush=True)
    return out


whiten.gram(X[:, :64], "cuda")                       # library load, first launches
t0 = time.perf_counter()
m = X[:, q].mean(axis=1, keepdims=True)
df = lap("host: X[:, q] - X[:, p]", lambda: X[:, q] - X[:, p])
S = lap("device: S = df df^T (H2D + mdx_gram_f64 + D2H)", lambda: whiten.gram(df, "cuda")) / df.shape[1]
L = lap("host: cholesky(S)", lambda: whiten.cholesky(S))
P = lap("host: inv(L)", lambda: np.linalg.inv(L))
df2 = lap("device: P (X - m) (H2D + mdx_project_f64 + D2H)", lambda: whiten.project(P, X, m, "cuda"))
Dm = lap("device: D = df df^T", lambda: whiten.gram(df2, "cuda"))
ev = lap("host: np.linalg.eig(D)  [the reference's call]", lambda: np.linalg.eig(Dm))
ev2 = lap("host: np.linalg.eigh(D)", lambda: np.linalg.eigh(Dm))
Dd = torch.from_numpy(Dm).cuda()
torch.linalg.eigh(Dd[:64, :64])
ev3 = lap("device: torch.linalg.eigh(D)", lambda: torch.linalg.eigh(Dd))
print("stage by stage with the host eig (the reference's statement) %9.1f ms" % ((time.perf_counter() - t0) * 1e3))
for _ in range(2):
    t1 = time.perf_counter(); m2, P2 = whiten.whitenlearn(X, q, p, device="cuda"); print("whiten.whitenlearn() (device-resident)          %9.1f ms" % ((time.perf_counter() - t1) * 1e3))
# the same whitening as the stage-by-stage host result: whitened dot products agree
order = ev[0].argsort()[::-1]
P_ref = np.dot(ev[1][:, order].T, P)
Y = X[:, :300] - m
a, b = P2 @ Y, np.real(P_ref) @ Y
print("max |whitened dot products - reference statement's| = %.2e (of magnitude %.2e)" % (np.abs(a.T @ a - b.T @ b).max(), np.abs(b.T @ b).max()))
t1 = time.perf_counter(); whiten.pcawhitenlearn(X, device="cuda"); print("whiten.pcawhitenlearn() (device-resident)       %9.1f ms" % ((time.perf_counter() - t1) * 1e3))
