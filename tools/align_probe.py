import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mdir_amd import ops
from oracle import chain as OC
rng = np.random.default_rng(0)
n, d, nq = 40001, 64, 21
db = (rng.standard_normal((n, d)) / 8).astype(np.float32)
qv = (rng.standard_normal((nq, d)) / 8).astype(np.float32)
want = OC.scores_chain(np.ascontiguousarray(db.T), np.ascontiguousarray(qv.T))
for off in (0, 1, 2, 3):
    base = torch.zeros(n * d + 8, device="cuda")
    view = base[off:off + n * d].view(n, d)
    view.copy_(torch.from_numpy(db))
    got = ops.scores_rowmajor(view, torch.from_numpy(qv).cuda(), "ND").cpu().numpy()
    print("offset %d floats (address %% 16 = %d): bit-exact %s" % (off, view.data_ptr() % 16, np.array_equal(got, want)), flush=True)
