"""Race screen for the loader/consumer similarity kernels (exact chain, fp16 shard, and -- round 4 -- MDX_F32_SPLIT3 within 2e-6): many random shapes, repeated launches,
every score compared bit for bit with the fmaf-chain oracle (a ring-protocol bug shows up as rare
wrong tiles that come and go with shape and load)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdir_amd import ops
from oracle import chain as OC

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
t0 = time.time()
# a background stream of work on another HIP stream makes timing uneven
bg = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device="cuda")
for it in range(iters):
    n = int(rng.integers(1, 90000)) if it % 3 else int(rng.integers(60000, 140000))
    d = int(rng.choice([32, 64, 100, 256, 512, 1000, 2048]))
    nq = int(rng.integers(1, 140))
    storage = "f16" if it % 5 == 4 else "f32"
    db = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
    q = (rng.standard_normal((nq, d)) / np.sqrt(d)).astype(np.float32)
    ix = ops.DescriptorIndex(torch.from_numpy(db).cuda(), "ND", storage=storage)
    qd = torch.from_numpy(q).cuda()
    if storage == "f32":
        want = OC.gemm_nt_chain(q, db)
    else:
        want = (q.astype(np.float16).astype(np.float64) @ db.astype(np.float16).astype(np.float64).T)
    for rep in range(4):
        with torch.cuda.stream(bg):
            junk2 = junk @ junk if rep % 2 else None
        got = ix.scores(qd, "ND").cpu().numpy()
        ok = np.array_equal(got, want) if storage == "f32" else np.allclose(got, want, rtol=0, atol=2e-6)
        if storage == "f32" and rep % 2 == 0:       # the labelled split-precision mode on the same index, between exact launches
            for mode in ("split3", "split2"):
                got3 = ix.scores(qd, "ND", compute=mode).cpu().numpy()
                if not (np.abs(got3 - want).max() <= 2e-6):
                    bad += 1
                    print("%s MISMATCH it=%d rep=%d n=%d d=%d nq=%d: max diff %.3g" % (mode, it, rep, n, d, nq, np.abs(got3 - want).max()))
        if storage == "f32" and rep % 2 == 1:       # the same rows read in place (mdx_scores_rowmajor): bit-exact too; both query layouts
            dbd = torch.from_numpy(db).cuda()
            got_rm = (ops.scores_rowmajor(dbd, qd, "ND") if rep == 1 else ops.scores_rowmajor(dbd, qd.t().contiguous(), "DN")).cpu().numpy()
            if not np.array_equal(got_rm, want):
                bad += 1
                print("rowmajor MISMATCH it=%d rep=%d n=%d d=%d nq=%d: %d wrong" % (it, rep, n, d, nq, int((got_rm != want).sum())))
        if storage == "f32" and nq <= 128 and rep == 3:      # round 6: the routed epilogue (mdx_scores_p2p, one rank: the buffer is local)
            p2p = ops.P2P(1, 0, nq, n, "cuda")
            p2p.connect([p2p.handle])
            ix.scores_p2p(qd, p2p, "ND")
            got_rt = p2p.close_step().cpu().numpy()
            p2p.close()
            if not np.array_equal(got_rt, want):
                bad += 1
                print("routed MISMATCH it=%d n=%d d=%d nq=%d: %d wrong" % (it, n, d, nq, int((got_rt != want).sum())))
        if not ok:
            bad += 1
            w = np.argwhere(got != want) if storage == "f32" else np.argwhere(np.abs(got - want) > 2e-6)
            print("MISMATCH it=%d rep=%d n=%d d=%d nq=%d %s: %d wrong, first %s" % (it, rep, n, d, nq, storage, len(w), w[:3].tolist()))
    ix.close()
print("stress done: %d iterations x4, %d mismatching launches, %.1f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
