"""Wall-clock of every stage of one evaluation (cirscore.py:54-71) at the headline size, descriptors already on the GPU:
index build, similarity, full ranking, mAP from the ranking (the reference's literal sequence, criterion `ranking: full`)
and mAP from rank positions (the default route).  What the bench's timed step does not show."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mdir_amd import ops
from mdir_amd.evaluate import compute_map_and_print, compute_map_and_print_from_scores

dev = torch.device("cuda", 0)
n = bench.N_ROXFORD + bench.N_DISTRACTORS
rows = bench.gen_rows(0, n, dev)
qvecs, qid = bench.gen_queries(n, dev)
gnd = bench.synth_gnd(bench.N_ROXFORD)
bench.plant_positives(rows, 0, n, gnd, qid, dev)
q_nd = qvecs.t().contiguous()


def lap(fn, reps=3):
    out, best = None, 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return out, best * 1e3


ix, t_build = lap(lambda: ops.DescriptorIndex(rows, "ND"))
sc, t_sc = lap(lambda: ix.scores(q_nd, "ND"))
sc_rm, t_rm = lap(lambda: ops.scores_rowmajor(rows, q_nd, "ND"))
assert torch.equal(sc, sc_rm)
rk, t_rk = lap(lambda: ops.rank_full(sc))
sink = io.StringIO()
with contextlib.redirect_stdout(sink):
    (a_full, _), t_map_full = lap(lambda: compute_map_and_print("roxford5k", rk.t(), gnd), reps=2)
    (a_pos, _), t_map_pos = lap(lambda: compute_map_and_print_from_scores("roxford5k", sc, gnd), reps=2)
assert a_full == a_pos, (a_full, a_pos)
print("index build %.2f ms | similarity %.2f | full ranking %.2f | mAP from the ranking %.2f | mAP from positions (no ranking) %.2f"
      % (t_build, t_sc, t_rk, t_map_full, t_map_pos))
print("literal route %.2f ms, default route %.2f ms; mAP-medium %.6f" % (t_build + t_sc + t_rk + t_map_full, t_build + t_sc + t_map_pos, a_full["map_medium"]))
print("the database read where it lies (mdx_scores_rowmajor, what score.py does for one evaluation): similarity %.2f ms, no build -> literal route %.2f ms, "
      "default route %.2f ms" % (t_rm, t_rm + t_rk + t_map_full, t_rm + t_map_pos))
