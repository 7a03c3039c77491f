import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
rng = np.random.default_rng(0)
D = 2048
A = rng.standard_normal((D, 3 * D)) * np.linspace(1.0, 0.05, D)[:, None]
M = A @ A.T
Md = torch.from_numpy(M).cuda()
for i in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    w, v = torch.linalg.eigh(Md)
    torch.cuda.synchronize(); print("torch.linalg.eigh on the GPU (f64, 2048): %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
t = time.perf_counter(); wh, vh = np.linalg.eigh(M); print("np.linalg.eigh: %.1f ms" % ((time.perf_counter() - t) * 1e3))
print("eigenvalues rel diff", float(np.abs(w.cpu().numpy() - wh).max() / wh.max()))
vd = v.cpu().numpy()
s = np.sign((vd * vh).sum(0))
print("eigenvectors max diff up to sign", float(np.abs(vd * s - vh).max()))
print("residual |M v - v w| device", float(np.abs(M @ vd - vd * w.cpu().numpy()).max() / wh.max()), "host", float(np.abs(M @ vh - vh * wh).max() / wh.max()))
