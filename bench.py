#!/usr/bin/env python3
"""Headline benchmark: queries/sec of the ranking hot path on the rOxford5k +
1M-distractor shaped workload (BASELINE.json configs[2]; configs[3] for --gpus N).

A "step" = one pass of the hot path over one batch of 70 queries against the
resident database: similarity (mdx_scores, fp32 MFMA) + exact full ranking
(mdx_rank_full), i.e. the reference's `np.dot(vecs.T, qvecs)` + `np.argsort(-scores,
axis=0)` (mdir/components/optim/score/cirscore.py:69-70).  Inputs are resident in
HBM when the timed region starts.  With --gpus N the 1M database is row-sharded
(strong scaling); see mdir_amd/sharded.py for the exchange.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows ROWS] [--no-cpu-baseline] [--no-secondary]
                    [--no-pipelined] [--extract-images M] [--profile] [--comm torch|mdx]

`--profile` = the headline loop alone (no CPU baseline, side legs, two-stream leg or extraction): the form
tools/profile_round.sh runs under rocprofv3, so that the per-kernel averages of the committed profile add up to the step.

Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

N_ROXFORD, N_DISTRACTORS = 4993, 1_000_000
NQ, DIM = 70, 2048
GEN_BLOCK = 4096
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0
# Two correct fp32 evaluations of one 2048-term dot product of unit vectors (BLAS order on the host, k-ordered fma
# chain on the GPU) differ by summation order only: <~ D * 2^-24 * |s| ~ 2e-6 for the |s| <= 0.02 of near-tied
# distractors (measured: 8e-8).  The CPU and GPU rankings may disagree only between scores closer than this -- 5x
# tighter than the north star's 1e-5 score tolerance, so that a real regression cannot hide under it.
SUM_ORDER_TOL = 2e-6


def gen_rows(lo, hi, device):
    """Rows [lo,hi) of the synthetic database as [hi-lo, D]: i.i.d. N(0,1), unit norm.
    Block-seeded so that any rank regenerates any row identically."""
    out = torch.empty((hi - lo, DIM), dtype=torch.float32, device=device)
    b = lo // GEN_BLOCK
    while b * GEN_BLOCK < hi:
        g = torch.Generator(device=device)
        g.manual_seed(1000 + b)
        blk = torch.randn((GEN_BLOCK, DIM), generator=g, device=device, dtype=torch.float32)
        blk /= blk.norm(dim=1, keepdim=True)
        s, e = max(lo, b * GEN_BLOCK), min(hi, (b + 1) * GEN_BLOCK)
        out[s - lo:e - lo] = blk[s - b * GEN_BLOCK:e - b * GEN_BLOCK]
        b += 1
    return out


def gen_queries(n_total, device):
    """70 queries = 70 distinct database rows + 0.05 N(0,1) noise, re-normalised ([D,Q])."""
    rng = np.random.default_rng(0)
    first = N_ROXFORD if n_total >= N_ROXFORD + NQ else 0      # keep query sources off the labelled rows
    qid = first + np.sort(rng.choice(n_total - first, size=NQ, replace=False))
    rows = torch.cat([gen_rows(int(i), int(i) + 1, device) for i in qid])
    g = torch.Generator(device=device)
    g.manual_seed(7)
    q = rows + 0.05 * torch.randn((NQ, DIM), generator=g, device=device)
    q /= q.norm(dim=1, keepdim=True)
    return q.t().contiguous(), qid


def synth_gnd(n_labelled):
    """rOxford-shaped ground truth: easy 5 / hard 10 / junk 5 disjoint random ids per
    query among the first 4993 rows."""
    rng = np.random.default_rng(1)
    gnd = []
    for _ in range(NQ):
        ids = rng.choice(n_labelled, size=20, replace=False)
        gnd.append({"easy": np.sort(ids[:5]), "hard": np.sort(ids[5:15]), "junk": np.sort(ids[15:]), "bbx": None})
    return gnd


def plant_positives(rows, lo, hi, gnd, qid, device):
    """Make the labelled rows relevant: row = normalise(source_row(q) + b * N(0,1)) with
    b = 0.03 (easy; cos to the query ~0.24), 0.1 (hard; ~0.09, inside the top distractors'
    range so that mAP < 1), 0.02 (junk, near duplicates).  Only rows in [lo,hi) are touched."""
    for q, g in enumerate(gnd):
        src = gen_rows(int(qid[q]), int(qid[q]) + 1, device)[0]
        for key, b in (("easy", 0.03), ("hard", 0.1), ("junk", 0.02)):
            for i in g[key]:
                i = int(i)
                if lo <= i < hi:
                    gen = torch.Generator(device=device)
                    gen.manual_seed(50_000_000 + i)
                    v = src + b * torch.randn(DIM, generator=gen, device=device)
                    rows[i - lo] = v / v.norm()


def cpu_baseline(vecs_dn_host, qvecs_host, reps=3):
    """The reference's two ranking statements through the numpy oracle, host cores."""
    from oracle import oracle as O
    dots, sorts = [], []
    for _ in range(reps):            # median of 3
        t0 = time.perf_counter()
        sc = O.scores(vecs_dn_host, qvecs_host)
        t1 = time.perf_counter()
        rk = np.argsort(-sc, axis=0)     # the reference's literal statement (default kind)
        t2 = time.perf_counter()
        dots.append(t1 - t0)
        sorts.append(t2 - t1)
    return sc, rk, float(np.median(dots)), float(np.median(sorts))


def cpu_dot_three_threads(vecs_dn_host, qvecs_host):
    """np.dot with the BLAS pool limited to 3 threads: what the reference's `OMP_NUM_THREADS=3`
    (mdir/stages/validate.py:10-12) gives when the BLAS honours it."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        return None
    from oracle import oracle as O
    with threadpool_limits(limits=3, user_api="blas"):
        t0 = time.perf_counter()
        O.scores(vecs_dn_host, qvecs_host)
        return time.perf_counter() - t0


def spread(values, digits=4):
    """min / median / max of the per-step figures of the timed region (>= 20 timed reps, median)."""
    v = np.asarray(values, dtype=np.float64)
    return {"min": round(float(v.min()), digits), "median": round(float(np.median(v)), digits), "max": round(float(v.max()), digits)}


def committed_traffic():
    """The newest committed PMC summary (profiles/rNN_traffic.json: FETCH_SIZE x2 + WRITE_SIZE per launch, separate rocprofv3
    --pmc passes, tools/profile_round.sh).  NOT measured in the bench run; a summary taken with other kernel sources is
    reported as stale and its figure dropped.  Returns (file's dict, similarity bytes per launch, ranking bytes, sources)."""
    tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not tf:
        return {}, None, None, None, None
    whole = json.load(open(tf[-1]))
    name = os.path.relpath(tf[-1], ROOT)
    src = "committed PMC summary %s (not measured in this run)" % name
    traffic, rank_traffic, rank_src = whole.get("scores_kernel_hbm_bytes_per_launch"), ranking_traffic(whole), src
    prof = whole
    if whole.get("scores_kernel_source_sha16") != scores_source_sha16():
        traffic, prof = None, {}
        src += ": STALE (similarity kernel sources changed since; re-profile with tools/profile_round.sh)"
    if whole.get("rank_source_sha16") not in (None, rank_source_sha16()):
        rank_traffic, rank_src = None, rank_src + ": STALE (mdx_rank.hip changed since)"
    return prof, traffic, src, rank_traffic, rank_src


def scores_source_sha16():
    """Hash of the similarity kernel's sources: ties a committed PMC figure to the kernel it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in ("mdx_scores_kernel.h", "mdx_index.hip"):
        with open(os.path.join(ROOT, "mdir_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def rank_source_sha16():
    """Hash of the ranking kernels' source: ties the committed PMC traffic of the sort to the code it was measured on."""
    import hashlib
    with open(os.path.join(ROOT, "mdir_amd", "csrc", "mdx_rank.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def ranking_traffic(prof):
    """HBM bytes of ONE full ranking from a committed PMC summary: the 4 histogram, 4 scan and 4 scatter launches of the
    LSD sort (per-launch figures of profiles/rNN_traffic.json; the scan kernel runs once per pass)."""
    if prof.get("ranking_hbm_bytes_per_ranking"):
        return prof["ranking_hbm_bytes_per_ranking"]
    total, seen = 0.0, 0
    for name, t in prof.get("per_kernel", {}).items():
        if "::sort_hist_kernel" in name or "::sort_scatter_kernel" in name:
            total, seen = total + t["total_bytes"], seen + 1
        elif "::sort_scan_kernel" in name:
            total, seen = total + 4 * t["total_bytes"], seen + 1
    return total if seen == 9 else None


def verify_ranking(sc, rk):
    """Device-side check of a full ranking at any size: every row of `rk` is a permutation of 0..n-1, scores are
    non-increasing along it, and ids ascend inside every run of equal scores (the tie rule) -- the properties a
    mis-ordered wave rank (ds_add_rtn form, csrc/mdx_rank.hip) would break."""
    nq, n = rk.shape
    ok_perm = ok_order = True
    for q in range(nq):
        seen = torch.zeros(n, dtype=torch.int32, device=rk.device)
        seen.index_add_(0, rk[q], torch.ones(n, dtype=torch.int32, device=rk.device))
        ok_perm = ok_perm and bool((seen == 1).all())
        s = sc[q][rk[q]]
        ok_order = ok_order and bool((s[:-1] >= s[1:]).all()) and bool(((s[:-1] != s[1:]) | (rk[q, :-1] < rk[q, 1:])).all())
    return ok_perm, ok_order


def side_legs(args, sharded, qvecs, sc, rk, ws, gnd, device, extra):
    """Side leg of the single-GPU run that launches the headline's similarity kernel with the sort-free evaluation (skipped by
    --no-pipelined / --profile).  The two-stream leg of rounds 2-4 (ranking of batch k beside the similarity of batch k+1) is
    gone: it measured +0.2 ... +1 %, and profiles/r05_overlap.md shows why -- the similarity kernel's 16 waves per CU hold every
    SIMD's whole register file, so the sort's workgroups only become resident when it ends."""
    from mdir_amd import ops
    # the same evaluation without materialising a ranking: similarity + rank positions of the labelled
    # ids (mdx_rank_of) -- what compute_map actually needs; identical mAP (asserted above), reported beside
    from mdir_amd.ops import _csr
    lists = [np.concatenate([g["easy"], g["hard"], g["junk"]]) for g in gnd]
    ids_t, off_t, _ = _csr(lists, device)
    cnt = torch.zeros(ids_t.numel(), dtype=torch.int64, device=device)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        sharded.index.scores(qvecs, "DN", out=sc)
        cnt.zero_()
        ops.rank_count_(cnt, sc, 0, ops.gather_scores(sc, ids_t, off_t), ids_t, off_t)
    torch.cuda.synchronize()
    t_pos = (time.perf_counter() - t1) / args.steps
    extra["sort_free_map_route"] = {"value": round(NQ / t_pos, 2), "unit": "queries/s", "ms_per_step": round(t_pos * 1e3, 4),
                                    "what": "similarity + rank positions of the %d labelled ids (no full ranking), same mAP"
                                            % int(ids_t.numel())}


def launch_ranks(n):
    """One child `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>`; returns its exit code.
    The children inherit stdout, so rank 0's JSON line is this command's output."""
    import socket
    import subprocess
    dryrun = os.environ.get("MDIR_AMD_DRYRUN_ONE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < n and not dryrun:
        print("bench.py --gpus %d: this node shows %d GPU(s) (MDIR_AMD_DRYRUN_ONE_GPU=1 runs all ranks on one GPU over "
              "gloo: a functional dry run, not a measurement)" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="n", type=int, default=N_ROXFORD + N_DISTRACTORS, help="database rows (default 1 004 993)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] / configs[4] side legs")
    ap.add_argument("--extract-images", type=int, default=40,
                    help="images PER SIZE (16 sizes) of the (untimed) descriptors/sec leg: ResNet101-GeM, 3 scales + whitening; 0 = skip")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the sort-free evaluation leg (name kept from the rounds that also had a "
                    "two-stream leg here)")
    ap.add_argument("--profile", action="store_true", help="the headline loop only: --no-cpu-baseline --no-secondary --no-pipelined --extract-images 0")
    ap.add_argument("--comm", choices=("torch", "mdx"), default=None,
                    help="N > 1: the exchange of partial scores through torch.distributed (default) or through the C-ABI communicator "
                         "(mdx_comm_* / mdx_exchange_scores over RCCL; same as MDIR_AMD_COMM=mdx)")
    args = ap.parse_args()
    if args.profile:
        args.no_cpu_baseline = args.no_secondary = args.no_pipelined = True
        args.extract_images = 0
    if args.comm:
        os.environ["MDIR_AMD_COMM"] = args.comm

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: start the N rank processes as CHILDREN (one per GPU, RCCL) and relay
        # rank 0's line and the exit code.  Decided here, before anything has touched the GPU (`device_count` does not
        # initialise it); this process never does, and nothing is exec'ed.
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # MDIR_AMD_DRYRUN_ONE_GPU=1: every rank uses cuda:0 and gloo (host-staged collectives) --
    # a functional dry run of the N>1 code path on a 1-GPU box, never a measurement.
    dryrun = os.environ.get("MDIR_AMD_DRYRUN_ONE_GPU") == "1"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from mdir_amd.sharded import private_miopen_caches
        private_miopen_caches(os.environ.get("LOCAL_RANK", "0"))      # before this process's first convolution (the extraction leg)
        import datetime
        limit = datetime.timedelta(seconds=900)       # a collective that cannot complete ends the run instead of hanging it
        if dryrun:
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=device, timeout=limit)
        # communicator set-up (lazy peer connections) is not a step: ShardedIndex runs a small all-to-all when it is
        # built and all ranks agree there (all-reduce) on the exchange form -- see mdir_amd/sharded.py

    from mdir_amd import ops
    from mdir_amd.sharded import ShardedIndex, shard_bounds
    from mdir_amd.evaluate import compute_map_and_print, compute_map_and_print_from_scores

    n_total = args.n
    lo, hi = shard_bounds(n_total, world, rank)
    rows = gen_rows(lo, hi, device)                 # [n_local, D]
    qvecs, qid = gen_queries(n_total, device)       # [D, Q]
    gnd = synth_gnd(min(N_ROXFORD, n_total))
    plant_positives(rows, lo, hi, gnd, qid, device)
    t0 = time.perf_counter()
    sharded = ShardedIndex(rows, "ND", n_total)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    n_local = hi - lo

    if world == 1:
        sc = torch.empty((NQ, n_total), dtype=torch.float32, device=device)
        rk = torch.empty((NQ, n_total), dtype=torch.int64, device=device)
        ws = torch.empty(ops.rank_workspace_bytes(n_total, NQ), dtype=torch.uint8, device=device)
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]

        def step(i=None):
            if i is not None:
                ev[i][0].record()
            sharded.index.scores(qvecs, "DN", out=sc)
            if i is not None:
                ev[i][1].record()
            ops.rank_full(sc, out=rk, workspace=ws)
            if i is not None:
                ev[i][2].record()
    else:
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(args.steps)]
        keep = {}

        sharded.reuse_buffers = True           # the loop keeps only the newest result: one receive buffer per chunk

        def step(i=None):
            if i is not None:
                ev[i][0].record()
            keep["rk"], keep["sc"], keep["q"] = sharded.rank_queries(qvecs, "DN")
            if i is not None:
                ev[i][1].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dryrun else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- results check + mAP (untimed) --------------------------------------
    extra = {}
    if world == 1:
        with contextlib.redirect_stdout(sys.stderr):      # stdout carries the one JSON line only
            avg_s, _ = compute_map_and_print_from_scores("roxford5k", sc, gnd)       # counting kernel, no sort
            avg_r, _ = compute_map_and_print("roxford5k", rk.t(), gnd)               # from the full ranking
        assert avg_s == avg_r, (avg_s, avg_r)
        perm_ok, order_ok = verify_ranking(sc, rk)
        assert perm_ok and order_ok, "the full ranking is not a stable descending permutation (perm %s, order %s)" % (perm_ok, order_ok)
        extra["ranking_verified_on_device"] = "all %d rows: permutation, non-increasing scores, ascending ids inside ties" % NQ
        assert bool((rk[:, 0].cpu() == torch.from_numpy(qid)).all()), "every query must retrieve its source row first"
        extra["map_medium"] = avg_r["map_medium"]
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in ev]))
        rank_ms = float(np.mean([b.elapsed_time(c) for _, b, c in ev]))          # the 12 launches of one full ranking, HIP events
        flops = 2.0 * NQ * n_total * DIM
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        algo_bytes = 4.0 * n_total * DIM + 4.0 * NQ * DIM + 4.0 * NQ * n_total
        # HBM traffic is NOT measured in this run: it is the per-launch PMC figure (FETCH_SIZE x2 + WRITE_SIZE, separate
        # rocprofv3 --pmc passes, tools/profile_round.sh) of the newest committed profile -- named in traffic_source
        # a profile of ANOTHER kernel says nothing about this one: the summary records the hash of the similarity
        # kernel's sources it was taken with (tools/summarize_profile.py); any other hash nulls the figure
        prof, traffic, traffic_source, rank_traffic, rank_src = committed_traffic()
        full_tiles, tail = divmod(NQ, 16)
        if n_total >= 32768 and 0 < tail <= 8 and full_tiles >= 1:
            kernel = "mdx::scores_lc_kernel<QT=%d,R=2,QR=1>: %d query tiles on v_mfma_f32_16x16x4 + the last %d queries on " \
                     "v_mfma_f32_4x4x1 (4 MFMA + 4 LDS-DMA loader waves)" % (full_tiles, full_tiles, tail)
        else:
            kernel = "mdx::scores_lc_kernel<QT=%d,R=%d> (fp32 MFMA 16x16x4; 4 MFMA + 4 LDS-DMA loader waves)" \
                     % (-(-NQ // 16), 2 if n_total >= 32768 else 1)
        roofline = {"kernel": kernel, "bound": "mfma",
                    "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel_ms": round(kernel_ms, 4), "algorithmic_flops": flops, "algorithmic_bytes": algo_bytes,
                    "hbm_GBps_at_algorithmic_bytes": round(algo_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                    "hbm_frac_of_8TBps": round(algo_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        if prof.get("scores_kernel_sustained_clock_ghz"):
            # from the same committed profile (SQ counter pass), not measured in this run: the kernel is power-limited, the peak
            # figure assumes 2.4 GHz (profiles/r02_scores_stamps.md)
            ghz = prof["scores_kernel_sustained_clock_ghz"]
            roofline["profiled_sustained_clock_ghz"] = ghz
            roofline["profiled_mfma_pipe_busy"] = prof.get("scores_kernel_mfma_pipe_busy")
            roofline["frac_of_peak_at_profiled_clock"] = round(achieved / (PEAK_F32_MFMA_TFLOPS * ghz / 2.4), 4)
        extra["roofline"] = roofline
        extra["rank_ms_per_step"] = round(rank_ms, 4)
        # how noisy THIS run was: the three HIP events of every timed step (one process's q/s swings +-2.5 % between processes
        # with the device's power state; the headline `value` stays steps / wall time)
        step_ms = [a.elapsed_time(c) for a, _, c in ev]
        extra["spread_over_timed_steps"] = {
            "kernel_ms": spread([a.elapsed_time(b) for a, b, _ in ev]), "rank_ms": spread([b.elapsed_time(c) for _, b, c in ev]),
            "step_ms": spread(step_ms), "value": spread([NQ / (t * 1e-3) for t in step_ms], 1), "steps": args.steps,
            "what": "HIP events on the launch stream around the similarity and the ranking of every timed step"}
        extra["step_ms_minus_kernels"] = round(elapsed / args.steps * 1e3 - kernel_ms - rank_ms, 4)     # host / launch gaps: ~0
        # the second kernel family of the step: np.argsort(-scores, axis=0) (cirscore.py:70) as a 4-pass LSD radix sort.
        # HBM-bound; algorithmic bytes = the argsort itself (4 B read + 8 B written per element), traffic = what
        # the four passes really move (committed PMC summary, like the similarity kernel's)
        rank_algo = 12.0 * NQ * n_total
        extra["roofline_rank"] = {
            "kernel": "mdx::sort_hist_kernel / sort_scan_kernel / sort_scatter_kernel x 4 passes (8-bit LSD radix, packed intermediates)",
            "bound": "hbm", "achieved": round(rank_algo / (rank_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(rank_algo / (rank_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": rank_traffic, "traffic_source": rank_src,
            "kernel_ms": round(rank_ms, 4), "algorithmic_bytes": rank_algo,
            "traffic_over_algorithmic": round(rank_traffic / rank_algo, 2) if rank_traffic else None,
            "hbm_GBps_at_real_traffic": round(rank_traffic / (rank_ms * 1e-3) / 1e9, 1) if rank_traffic else None,
            "what": "12 launches per ranking; the 4-pass form moves ~4.8x the bytes of an ideal one-pass argsort and streams them at the "
                    "rate HBM gives a read+write mix (DESIGN section 4): only fewer passes would help, and the MSD / one-sweep forms measured slower"}
        if not args.no_pipelined:
            side_legs(args, sharded, qvecs, sc, rk, ws, gnd, device, extra)
    else:
        rk_mine, sc_mine, (qlo, qhi) = keep["rk"], keep["sc"], keep["q"]
        ok = torch.tensor([1], device="cpu" if dryrun else device)
        if qhi > qlo:
            ok[0] = int(bool((rk_mine[:, 0].cpu() == torch.from_numpy(qid[qlo:qhi])).all()))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        assert int(ok.item()) == 1, "sharded ranking lost a query's source row"
        # every rank's rows of the global ranking, checked on its device against the exchanged scores: permutations of
        # 0..N-1 (global ids), non-increasing, ascending ids inside ties -- what the single-GPU line asserts for all 70
        if qhi > qlo:
            perm_ok, order_ok = verify_ranking(sc_mine.dense(), rk_mine)
            ok[0] = int(perm_ok and order_ok)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        assert int(ok.item()) == 1, "a rank's rows of the sharded ranking are not stable descending permutations"
        extra["ranking_verified_on_device"] = "every rank's query rows: permutation of the global ids, non-increasing scores, ascending ids inside ties"
        # mAP without any ranking: counting kernel + two tiny all-reduces
        s_local = sharded.local_scores(qvecs, "DN")
        from mdir_amd.evaluate import map_from_positions
        oks = [np.concatenate([g["easy"], g["hard"]]) for g in gnd]
        junks = [g["junk"] for g in gnd]
        pos, off = sharded.positions(s_local, [np.concatenate([o, j]) for o, j in zip(oks, junks)])
        pos = pos.cpu().numpy()
        pl = [pos[off[q]:off[q] + len(oks[q])] for q in range(NQ)]
        jl = [pos[off[q] + len(oks[q]):off[q + 1]] for q in range(NQ)]
        extra["map_medium"] = map_from_positions(pl, jl, [len(o) for o in oks])[0]
        # per-phase breakdown of the LAST timed step on every rank (HIP events on the compute stream) and a head count
        ph = sharded.phase_ms() or {"scores_ms": float("nan"), "exchange_exposed_ms": float("nan"), "sort_ms": float("nan")}
        mine = torch.tensor([ph["scores_ms"], ph["exchange_exposed_ms"], ph["sort_ms"], 1.0], dtype=torch.float64,
                            device="cpu" if dryrun else device)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        table = torch.stack(gathered).cpu().numpy()
        extra["nranks_seen"] = int(round(float(table[:, 3].sum())))
        # the same roofline objects as the single-GPU line, per rank: a rank multiplies its shard (2 Q n_local D flop on the
        # fp32 MFMA) and sorts its queries' rows of the whole database (12 B per element of [Q_mine, N]); the job's figure is
        # the SLOWEST rank's (the step waits for it)
        from mdir_amd.sharded import shard_bounds as _sb
        per_rank_tf, per_rank_gbs = [], []
        for r in range(world):
            rl, rh = _sb(n_total, world, r)
            qb = (NQ // world) + (1 if r < NQ % world else 0)
            t_s, t_r = float(table[r, 0]), float(table[r, 2])
            per_rank_tf.append(round(2.0 * NQ * (rh - rl) * DIM / (t_s * 1e-3) / 1e12, 2) if t_s > 0 else None)
            per_rank_gbs.append(round(12.0 * qb * n_total / (t_r * 1e-3) / 1e9, 1) if t_r > 0 and qb else None)
        tf_ok = [x for x in per_rank_tf if x]
        # HBM traffic per rank: the committed single-GPU PMC figure scaled by the shard's share of the rows (the kernel streams
        # its rows once, the traffic is linear in them: 1.007x algorithmic at N = 1 M) -- derived, labelled, not measured here
        _, t1, t1_src, r1, r1_src = committed_traffic()
        n_big = max(_sb(n_total, world, r)[1] - _sb(n_total, world, r)[0] for r in range(world))
        q_big = -(-NQ // world)
        n_prof = N_ROXFORD + N_DISTRACTORS
        traffic_rank = round(t1 * n_big / n_prof, 1) if t1 else None
        traffic_sort = round(r1 * (q_big * n_total) / (NQ * n_prof), 1) if r1 else None
        if tf_ok:
            extra["roofline"] = {"kernel": "mdx::scores_lc_kernel (fp32 MFMA 16x16x4 [+ 4x4x1 leftover]; 4 MFMA + 4 LDS-DMA loader waves), per rank on its shard",
                                 "bound": "mfma", "achieved": min(tf_ok), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s per GPU (slowest rank)",
                                 "frac": round(min(tf_ok) / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic_rank,
                                 "traffic_source": (t1_src + "; x %d / %d rows: the largest shard's share of the single-GPU launch" % (n_big, n_prof)) if t1_src else None,
                                 "algorithmic_flops_per_rank": 2.0 * NQ * n_big * DIM,
                                 "algorithmic_bytes_per_rank": 4.0 * n_big * DIM + 4.0 * NQ * DIM + 4.0 * NQ * n_big,
                                 "per_rank_achieved": per_rank_tf,
                                 "kernel_ms_per_rank": [round(float(x), 4) for x in table[:, 0]],
                                 "what": "HIP events on each rank's compute stream around the similarity kernels of the last timed step "
                                         "(chunked shards: the sum of the chunks' launches)"}
        gb_ok = [x for x in per_rank_gbs if x]
        if gb_ok:
            extra["roofline_rank"] = {"kernel": "mdx::sort_* x 4 passes over the peer blocks (mdx_rank_full_segments), per rank on its queries",
                                      "bound": "hbm", "achieved": min(gb_ok), "peak": PEAK_HBM_GBS, "unit": "GB/s per GPU (slowest rank)",
                                      "frac": round(min(gb_ok) / PEAK_HBM_GBS, 4), "traffic": traffic_sort,
                                      "traffic_source": (r1_src + "; x (%d x %d) / (%d x %d) elements" % (q_big, n_total, NQ, n_prof)) if r1_src else None,
                                      "per_rank_achieved": per_rank_gbs,
                                      "algorithmic_bytes_per_rank": [12.0 * ((NQ // world) + (1 if r < NQ % world else 0)) * n_total for r in range(world)]}
        step_ms = [a.elapsed_time(b) for a, b in ev]
        extra["spread_over_timed_steps"] = {"step_ms": spread(step_ms), "value": spread([NQ / (t * 1e-3) for t in step_ms], 1), "steps": args.steps,
                                            "what": "rank 0: HIP events on its compute stream around every timed step (similarity, exchange wait, sort)"}
        extra["comm"] = "mdx (C ABI: mdx_exchange_scores over RCCL)" if getattr(sharded, "_comm", None) is not None else "torch.distributed"
        extra["phases_ms_per_rank"] = {"scores": [round(float(x), 4) for x in table[:, 0]],
                                       "exchange_exposed": [round(float(x), 4) for x in table[:, 1]],
                                       "sort": [round(float(x), 4) for x in table[:, 2]],
                                       "exchange": "all_to_all" if sharded._use_a2a else "all_gather", "chunks": sharded.chunks,
                                       "what": "last timed step; exchange_exposed = compute-stream wait for transfers after the last "
                                               "similarity kernel (+ re-block copy)"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            vecs_host = rows.t().contiguous().cpu().numpy()       # reference layout [D,N]
            _, rk_cpu, t_dot, t_sort = cpu_baseline(vecs_host, qvecs.cpu().numpy())
            import multiprocessing
            extra["cpu_baseline"] = {
                "value": round(NQ / (t_dot + t_sort), 3), "unit": "queries/s",
                "cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else multiprocessing.cpu_count(),
                "kind": "port",
                "sample": "full workload, median of 3: np.dot %.2f s (BLAS, all cores) + np.argsort %.2f s (1 thread), "
                          "N=%d Q=%d D=%d fp32" % (t_dot, t_sort, n_total, NQ, DIM)}
            t_dot3 = cpu_dot_three_threads(vecs_host, qvecs.cpu().numpy())
            if t_dot3 is not None:
                extra["cpu_baseline"]["value_blas_3_threads"] = round(NQ / (t_dot3 + t_sort), 3)
                extra["cpu_baseline"]["sample"] += "; with the BLAS pool at 3 threads (the reference's OMP_NUM_THREADS=3) np.dot takes %.2f s" % t_dot3
            # parity with the reference CPU path at full size: the two statements differ only in the summation order of
            # the 2048-term dot products (BLAS vs the k-ordered chain), i.e. in the last bits of near-tied scores
            with contextlib.redirect_stdout(sys.stderr):
                avg_cpu, _ = compute_map_and_print("roxford5k", rk_cpu, gnd)
            extra["map_medium_cpu"] = avg_cpu["map_medium"]
            gpu_top = rk[:, :100].t().cpu().numpy()
            differ = np.argwhere(rk_cpu[:100] != gpu_top)                       # (slot, query)
            extra["cpu_top100_id_agreement"] = round(1.0 - len(differ) / gpu_top.size, 6)
            max_gap = 0.0
            if len(differ):
                qs = torch.from_numpy(differ[:, 1]).to(device)
                a = sc[qs, torch.from_numpy(gpu_top[differ[:, 0], differ[:, 1]]).to(device)]
                b = sc[qs, torch.from_numpy(rk_cpu[:100][differ[:, 0], differ[:, 1]]).to(device)]
                max_gap = float((a - b).abs().max())
            extra["cpu_top100_max_score_gap_where_ids_differ"] = max_gap
            # ids may only differ between scores closer than the summation-order bound (north-star tolerance: 1e-5)
            assert max_gap <= SUM_ORDER_TOL, "CPU and GPU rankings differ between scores %.3g apart" % max_gap
            # positions of the labelled rows (all that mAP depends on) under both rankings; a row may sit elsewhere only
            # if its GPU score has a neighbour in the GPU ranking closer than the score tolerance (a near-tie)
            labelled_pos_equal, worst, n_moved = True, 0.0, 0
            inv_cpu = np.empty(n_total, dtype=np.int64)
            for q in range(NQ):
                ids = np.concatenate([gnd[q]["easy"], gnd[q]["hard"], gnd[q]["junk"]]).astype(np.int64)
                inv_cpu[rk_cpu[:, q]] = np.arange(n_total)
                ids_d = torch.from_numpy(ids).to(device)
                pos_gpu = torch.nonzero(rk[q].unsqueeze(0) == ids_d.unsqueeze(1))[:, 1].cpu().numpy()     # aligned with ids
                moved = np.nonzero(pos_gpu != inv_cpu[ids])[0]
                if len(moved):
                    labelled_pos_equal = False
                    n_moved += len(moved)
                    at = torch.from_numpy(np.clip(pos_gpu[moved], 1, n_total - 2)).to(device)
                    s0, sm, sp = sc[q, rk[q, at]], sc[q, rk[q, at - 1]], sc[q, rk[q, at + 1]]
                    worst = max(worst, float(torch.minimum((s0 - sm).abs(), (s0 - sp).abs()).max()))
            assert worst <= SUM_ORDER_TOL, "a labelled row ranks differently on the CPU path without a near-tie (gap %.3g)" % worst
            if labelled_pos_equal:
                assert avg_cpu["map_medium"] == extra["map_medium"], (avg_cpu["map_medium"], extra["map_medium"])
            extra["map_equals_cpu_path"] = bool(avg_cpu["map_medium"] == extra["map_medium"])
            extra["labelled_positions_equal_cpu_path"] = labelled_pos_equal
            extra["cpu_path_parity"] = {
                "labelled_rows_ranked_elsewhere": n_moved, "of": 20 * NQ, "their_gap_to_a_neighbouring_score": worst,
                "asserted_bound": SUM_ORDER_TOL, "top100_slots_with_other_ids": int(len(differ)),
                "what": "the CPU path (np.dot in BLAS order, numpy's unstable argsort) and the GPU path (k-ordered fma chain, "
                        "ties by ascending id) may order rows differently only inside runs of scores closer than the summation-order "
                        "bound 2e-6 (5x tighter than the 1e-5 score tolerance); asserted above for every such row.  mAP then differs by what such swaps of labelled rows move (compare map_medium "
                        "with map_medium_cpu); the printed 2-decimal mAP is the same"}
            del vecs_host, rk_cpu
        except Exception as exc:          # the reported baseline must not cost the measured line
            extra["cpu_baseline"] = {"value": None, "unit": "queries/s", "error": "%s: %s" % (type(exc).__name__, exc)}

    if world > 1 and not args.no_cpu_baseline:
        # The CPU reference beside an N > 1 line too (a SCALE line without it reads as unmeasured): after the timed region rank
        # 0 regenerates the WHOLE database (block-seeded rows: the same bits the shards hold), runs the reference's two
        # statements ONCE on the host cores (bounded: one repetition instead of the single-GPU line's median of 3) and checks the
        # job's mAP against the CPU path's; the other ranks wait at the barrier (their host threads idle).
        if rank == 0:
            try:
                full = gen_rows(0, n_total, device)
                plant_positives(full, 0, n_total, gnd, qid, device)
                vecs_host = full.t().contiguous().cpu().numpy()       # reference layout [D,N]
                del full
                _, rk_cpu, t_dot, t_sort = cpu_baseline(vecs_host, qvecs.cpu().numpy(), reps=1)
                import multiprocessing
                with contextlib.redirect_stdout(sys.stderr):
                    avg_cpu, _ = compute_map_and_print("roxford5k", rk_cpu, gnd)
                extra["cpu_baseline"] = {
                    "value": round(NQ / (t_dot + t_sort), 3), "unit": "queries/s",
                    "cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else multiprocessing.cpu_count(),
                    "kind": "port",
                    "sample": "full workload, ONE repetition on rank 0's host cores while the other %d ranks wait: np.dot %.2f s (BLAS, all "
                              "cores) + np.argsort %.2f s (1 thread), N=%d Q=%d D=%d fp32" % (world - 1, t_dot, t_sort, n_total, NQ, DIM)}
                extra["map_medium_cpu"] = avg_cpu["map_medium"]
                # same statement as the single-GPU line: BLAS order vs the k-ordered chain may swap rows inside near-ties only
                # (recorded, not asserted: an exception on rank 0 alone would leave the other ranks in the barrier below)
                extra["map_equals_cpu_path"] = bool(avg_cpu["map_medium"] == extra["map_medium"])
                extra["map_within_1e-5_of_cpu_path"] = bool(abs(avg_cpu["map_medium"] - extra["map_medium"]) <= 1e-5)
                # rank 0's queries: the head of its rows of the sharded ranking against the CPU ranking's columns
                rk0, (q0lo, q0hi) = keep["rk"], keep["q"]
                if q0hi > q0lo:
                    head = rk0[:, :100].t().cpu().numpy()
                    extra["cpu_top100_id_agreement"] = round(float((rk_cpu[:100, q0lo:q0hi] == head).mean()), 6)
                del vecs_host, rk_cpu
            except Exception as exc:          # the reported baseline must not cost the measured line
                extra["cpu_baseline"] = {"value": None, "unit": "queries/s", "error": "%s: %s" % (type(exc).__name__, exc)}
        dist.barrier()

    if rank == 0 and world == 1 and not args.no_secondary:
        # BASELINE.json's other single-GPU configurations, timed beside the headline (side legs: they cannot cost the
        # measured line): configs[1] rOxford5k alone (70 x 4 993, latency-bound) and configs[4]'s fp16 descriptors on the
        # fp16 MFMA (same 1 M x 2048 problem, shard stored as fp16: HBM-bound)
        try:
            def timed(fn, reps=20):
                for _ in range(3):
                    fn()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    fn()
                b.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b) / reps
            sec = {}
            small = ops.DescriptorIndex(rows[:N_ROXFORD].contiguous(), "ND")
            sc5 = torch.empty((NQ, N_ROXFORD), dtype=torch.float32, device=device)
            rk5 = torch.empty((NQ, N_ROXFORD), dtype=torch.int64, device=device)
            ws5 = torch.empty(ops.rank_workspace_bytes(N_ROXFORD, NQ), dtype=torch.uint8, device=device)
            t_s, t_r = timed(lambda: small.scores(qvecs, "DN", out=sc5)), timed(lambda: ops.rank_full(sc5, out=rk5, workspace=ws5))
            sec["configs1_roxford5k"] = {"workload": "N=%d Q=%d D=%d fp32, similarity + exact full ranking" % (N_ROXFORD, NQ, DIM),
                                         "scores_us": round(1e3 * t_s, 1), "rank_us": round(1e3 * t_r, 1),
                                         "queries_per_s": round(NQ / ((t_s + t_r) * 1e-3), 1), "bound": "launch latency"}
            small.close()
            # the serving form of configs[2]: the 100 best rows per query instead of the full ranking (mdx_topk, exact)
            t_k = timed(lambda: ops.topk(sc, 100, workspace=ws), reps=10)
            ids100, _ = ops.topk(sc, 100, workspace=ws)
            assert bool((ids100 == rk[:, :100]).all())                   # = the head of the full ranking
            kms = extra.get("roofline", {}).get("kernel_ms") or 0.0
            sec["configs2_top100"] = {"workload": "N=%d Q=%d: exact top-100 per query instead of the full ranking" % (n_total, NQ),
                                      "topk_ms": round(t_k, 4), "queries_per_s_with_the_fp32_similarity": round(NQ / ((kms + t_k) * 1e-3), 1) if kms else None}
            # one evaluation multiplies its database once: the same exact product on the row-major [N,D] matrix read where it lies
            # (mdx_scores_rowmajor: no index, no second 8 GB), next to what building an index for one product costs
            sc_rm = torch.empty_like(sc)
            t_rm = timed(lambda: ops.scores_rowmajor(rows, qvecs, "DN", out=sc_rm), reps=10)

            def build_multiply():
                ix1 = ops.DescriptorIndex(rows, "ND")
                ix1.scores(qvecs, "DN", out=sc_rm)
                ix1.close()
            t_bm = timed(build_multiply, reps=5)
            sec["configs2_one_evaluation"] = {
                "workload": "N=%d Q=%d D=%d fp32: the exact similarity of ONE evaluation, descriptors row-major on the device" % (n_total, NQ, DIM),
                "in_place_ms": round(t_rm, 4), "index_build_plus_multiply_ms": round(t_bm, 4), "resident_index_ms": kms or None,
                "bit_identical_to_the_index_route": bool(torch.equal(sc_rm, sc)),
                "what": "mdx_scores_rowmajor (the kernels of the headline on the caller's matrix) against mdx_index_create_in + mdx_scores + "
                        "destroy with the tiles in PyTorch's pool (own hipMalloc + hipFree of the 8 GB shard: ~190 ms per pair)"}
            assert sec["configs2_one_evaluation"]["bit_identical_to_the_index_route"]

            def wall(fn, reps=3):
                best = None
                for _ in range(reps):
                    torch.cuda.synchronize()
                    t_w = time.perf_counter()
                    with contextlib.redirect_stdout(sys.stderr):
                        out_w = fn()
                    torch.cuda.synchronize()
                    t_w = time.perf_counter() - t_w
                    best = t_w if best is None or t_w < best else best
                return out_w, best * 1e3
            # the whole evaluation of cirscore.py:65-71 on resident descriptors, wall clock: product (in place) + mAP
            (avg_d, _), t_default = wall(lambda: compute_map_and_print_from_scores("roxford5k", ops.scores_rowmajor(rows, qvecs, "DN", out=sc_rm), gnd))
            (avg_l, _), t_literal = wall(lambda: compute_map_and_print("roxford5k", ops.rank_full(ops.scores_rowmajor(rows, qvecs, "DN", out=sc_rm), out=rk, workspace=ws).t(), gnd))
            assert avg_d["map_medium"] == avg_l["map_medium"] == extra["map_medium"]
            sec["configs2_one_evaluation"].update({
                "evaluation_ms_default_route": round(t_default, 3), "evaluation_ms_literal_route": round(t_literal, 3),
                "routes": "default = product + rank positions of the labelled ids (mdx_rank_of) + host AP; literal = product + full argsort + "
                          "positions inside the ranking (mdx_rank_positions) + host AP; same mAP as the headline's"})
            del sc_rm
            # the LABELLED split-precision modes on the SAME fp32 shard (not the headline, not the parity contract -- timed beside it
            # with what they do to the result): MDX_F32_SPLIT3 = three bf16 pieces per operand, six products on the bf16 MFMA;
            # MDX_F32_SPLIT2 = block floating point, two fp16 pieces with a scaled residual, three products on the fp16 MFMA
            sc3 = torch.empty_like(sc)
            for mode, what, kern in (
                    ("split3", "MDX_F32_SPLIT3: x = h + m + l in bf16, products hh+hm+mh+hl+lh+mm on v_mfma_f32_16x16x32_bf16, fp32 accumulation",
                     "mdx::scores_split3_kernel<QT=5,R=2,NSTAGE=3,CW=8> (8 MFMA waves splitting in registers + 4 LDS-DMA loader waves)"),
                    ("split2", "MDX_F32_SPLIT2: block floating point, X = h + m / 2^11 in fp16 (scaled residual), products hh + (hm+mh) / 2^11 on "
                               "v_mfma_f32_16x16x32_f16, cross terms in their own accumulator",
                     "mdx::scores_split2_kernel<QT=5,R=2,NSTAGE=3,CW=8> (same ring; half the matrix work of split3)")):
                t_3 = timed(lambda: sharded.index.scores(qvecs, "DN", out=sc3, compute=mode), reps=10)
                b3 = 4.0 * n_total * DIM + 4.0 * NQ * n_total + (6.0 if mode == "split3" else 4.0) * 80 * DIM
                d3 = (sc3 - sc).abs()
                ids3, _ = ops.topk(sc3, 100, workspace=ws)
                with contextlib.redirect_stdout(sys.stderr):
                    avg3, _ = compute_map_and_print_from_scores("roxford5k", sc3, gnd)
                diff3 = torch.nonzero(ids3 != rk[:, :100])
                gap3 = 0.0
                if len(diff3):          # where the two top-100 lists name other rows: how far apart are those rows' EXACT scores?
                    qq = diff3[:, 0]
                    gap3 = float((sc[qq, ids3[qq, diff3[:, 1]]] - sc[qq, rk[qq, diff3[:, 1]]]).abs().max())
                assert float(d3.max()) <= SUM_ORDER_TOL and gap3 <= SUM_ORDER_TOL, (mode, float(d3.max()), gap3)
                sec[mode] = {
                    "workload": "N=%d Q=%d D=%d, the SAME fp32 shard, %s (labelled second mode; the exact chain stays the headline)" % (n_total, NQ, DIM, what),
                    "scores_ms": round(t_3, 4), "exact_chain_scores_ms": kms or None,
                    "roofline": {"kernel": kern, "bound": "hbm", "achieved": round(b3 / (t_3 * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": round(b3 / (t_3 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "algorithmic_bytes": b3, "traffic": None,
                                 "chain_equivalent_TFLOPs": round(2.0 * NQ * n_total * DIM / (t_3 * 1e-3) / 1e12, 1),
                                 "what": ("power-bound with real operands (all-zero operands: the stream-only time of the same kernel)" if mode == "split3"
                                          else "at the ring's stream-only time: half of split3's matrix work fits under the stream") + ", profiles/r04_split3.md"},
                    "queries_per_s_with_the_fp32_ranking": round(NQ / ((t_3 + extra.get("rank_ms_per_step", 0.0)) * 1e-3), 1),
                    "max_abs_diff_vs_exact_chain": float(d3.max()), "mean_abs_diff_vs_exact_chain": float(d3.mean()), "asserted_bound": SUM_ORDER_TOL,
                    "map_medium_" + mode: avg3["map_medium"], "map_medium_exact": extra.get("map_medium"),
                    "top100_slot_agreement_with_exact": round(1.0 - len(diff3) / ids3.numel(), 6),
                    "top100_max_exact_score_gap_where_ids_differ": gap3,
                    "top1_agreement_with_exact": round(float((ids3[:, 0] == rk[:, 0]).float().mean()), 6)}
            del sc3, d3
            half = ops.DescriptorIndex(rows, "ND", storage="f16")
            t_h = timed(lambda: half.scores(qvecs, "DN", out=sc), reps=10)
            hb = half.device_bytes + 4 * NQ * n_total
            sec["configs4_fp16_shard"] = {"workload": "N=%d Q=%d D=%d, shard and queries stored as fp16, v_mfma_f32_16x16x32_f16, fp32 accumulation"
                                                      % (n_total, NQ, DIM), "scores_ms": round(t_h, 4),
                                          "roofline": {"bound": "hbm", "achieved": round(hb / (t_h * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                                       "frac": round(hb / (t_h * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_bytes": float(hb)},
                                          "queries_per_s_with_the_fp32_ranking": round(NQ / ((t_h + extra.get("rank_ms_per_step", 0.0)) * 1e-3), 1),
                                          "contract": "scores within 2e-3 of fp32 (input rounding), tests/test_gpu_f16.py"}
            # what the fp16 shard does to the RESULT at this size (sc now holds the fp16-shard scores, rk the fp32 ranking)
            with contextlib.redirect_stdout(sys.stderr):
                avg16, _ = compute_map_and_print_from_scores("roxford5k", sc, gnd)
            ids16, _ = ops.topk(sc, 100, workspace=ws)
            sec["configs4_fp16_shard"].update({
                "map_medium_fp16": avg16["map_medium"], "map_medium_fp32": extra.get("map_medium"),
                "top100_slot_agreement_with_fp32": round(float((ids16 == rk[:, :100]).float().mean()), 6),
                "top1_agreement_with_fp32": round(float((ids16[:, 0] == rk[:, 0]).float().mean()), 6)})
            half.close()
            # configs[4]'s own shape: 247tokyo1k, query == database (1 125 x 1 125), VGG16 descriptors (512-d) stored as fp16
            g4 = torch.Generator(device=device)
            g4.manual_seed(4)
            tk = torch.randn((1125, 512), generator=g4, device=device)
            tk /= tk.norm(dim=1, keepdim=True)
            tix = ops.DescriptorIndex(tk, "ND", storage="f16")
            tsc = torch.empty((1125, 1125), dtype=torch.float32, device=device)
            trk = torch.empty((1125, 1125), dtype=torch.int64, device=device)
            tws = torch.empty(ops.rank_workspace_bytes(1125, 1125), dtype=torch.uint8, device=device)
            tq = tk.t().contiguous()
            t_s, t_r = timed(lambda: tix.scores(tq, "DN", out=tsc)), timed(lambda: ops.rank_full(tsc, out=trk, workspace=tws))
            assert bool((trk[:, 0] == torch.arange(1125, device=device)).all())          # every image retrieves itself first
            sec["configs4_247tokyo1k_shape"] = {"workload": "N=Q=1125 D=512, fp16 shard, query == database, similarity + exact full ranking",
                                                "scores_us": round(1e3 * t_s, 1), "rank_us": round(1e3 * t_r, 1),
                                                "queries_per_s": round(1125 / ((t_s + t_r) * 1e-3), 1), "bound": "launch latency"}
            tix.close()
            # rows f3 / f4 of SURVEY.md section 8: the float64 products of whitening learning (whiten.py:22,42,45,46) at
            # D = 2048 on 20 000 descriptors, and the CLAHE networks' input conversion on a batch of four 1024 x 768 images
            g5 = torch.Generator(device=device)
            g5.manual_seed(5)
            A64 = torch.randn((DIM, 20000), generator=g5, device=device, dtype=torch.float64)
            P64 = torch.randn((DIM, DIM), generator=g5, device=device, dtype=torch.float64)
            m64 = torch.randn(DIM, generator=g5, device=device, dtype=torch.float64)
            t_g, t_p = timed(lambda: ops.gram_f64(A64), reps=5), timed(lambda: ops.project_f64(P64, A64, m64), reps=5)
            tri = (DIM // 128) * (DIM // 128 + 1) // 2
            fl_g, fl_p = 2.0 * 20000 * 128 * 128 * tri, 2.0 * DIM * DIM * 20000
            sec["whitening_learning_f64"] = {
                "workload": "D=%d, n=20000 float64: mdx_gram_f64 (np.dot(df, df.T)) and mdx_project_f64 (np.dot(P, X-m))" % DIM,
                "gram_ms": round(t_g, 3), "project_ms": round(t_p, 3),
                "roofline_gram": {"bound": "mfma", "achieved": round(fl_g / t_g / 1e9, 2), "peak": 78.6, "unit": "TFLOP/s",
                                  "frac": round(fl_g / t_g / 1e9 / 78.6, 4), "what": "flops executed: upper-triangle tiles only"},
                "roofline_project": {"bound": "mfma", "achieved": round(fl_p / t_p / 1e9, 2), "peak": 78.6, "unit": "TFLOP/s",
                                     "frac": round(fl_p / t_p / 1e9 / 78.6, 4)}}
            del A64, P64
            u8 = torch.randint(0, 256, (4, 768, 1024, 3), generator=g5, device=device, dtype=torch.uint8)
            mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
            t_c, t_n = timed(lambda: ops.clahe_u8_to_chw(u8, 4, 8, mean, std)), timed(lambda: ops.u8_to_chw(u8, mean, std))
            cb = 4 * 768 * 1024 * (3 + 1 + 8 + 1 + 8 + 1 + 12)      # rgb in; L8 and chroma (a, b) written, then read; L8' and fp32 CHW out
            sec["clahe_preprocess"] = {
                "workload": "4 x 1024x768 uint8 RGB -> CLAHE (clip 4, 8x8 tiles) on the Lab lightness -> normalised fp32 CHW "
                            "(parity unpinned: OpenCV's algorithm restated)",
                "ms_per_batch": round(t_c, 4), "plain_u8_to_chw_ms_per_batch": round(t_n, 4),
                "roofline": {"bound": "hbm", "achieved": round(cb / (t_c * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                             "frac": round(cb / (t_c * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_bytes": float(cb)}}
            extra["secondary_configs"] = sec
        except Exception as exc:
            extra["secondary_configs"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if args.extract_images > 0:
        # second half of BASELINE.json's metric: descriptors/sec (every rank extracts its own images).  The number is
        # taken on an image LIST -- 16 JPEG sizes through the real loader -- in steady state; what a new size costs
        # and the resident single-shape figure of round 1 are reported beside it.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import types
        from bench_extract import measure, measure_list
        ex, err = None, None
        try:
            with contextlib.redirect_stdout(sys.stderr):
                ex = measure_list("resnet101", workers=8, short=12, mid=max(8, args.extract_images // 2), long=max(16, args.extract_images))
                ex["resident_single_shape"] = measure(types.SimpleNamespace(arch="resnet101", images=24, channels_last=False,
                                                                            miopen_find=False, batch=8))
                if world == 1:      # configs[4]'s network: VGG16-GeM, 3 scales + learned whitening, one resident 1024x768 shape
                    vg = measure(types.SimpleNamespace(arch="vgg16", images=24, channels_last=False, miopen_find=False, batch=8))
                    ex["vgg16_resident_single_shape_descriptors_per_s"] = vg["value"]
        except Exception as exc:        # an untimed side leg must not cost the ranking result (or hang the other ranks)
            err = "%s: %s" % (type(exc).__name__, exc)
        agg = torch.tensor([ex["value"] if ex else 0.0, 1.0 if ex else 0.0], dtype=torch.float64,
                           device="cpu" if dryrun else device)
        if world > 1:
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        if ex and int(agg[1].item()) == world:
            rs = ex.pop("resident_single_shape")
            extra["descriptors_per_s"] = dict(ex, value=round(float(agg[0].item()), 2), n_gpus=world,
                                              config="ResNet101-GeM random init, 3 scales + learned whitening through the wrapper "
                                                     "chain, fp32; JPEG files of 16 sizes (longer side 1024) through the loader",
                                              tail_ms_per_image_mdx=rs["tail_ms_per_image_mdx"],
                                              tail_ms_per_image_torch_ops=rs["tail_ms_per_image_torch_ops"],
                                              roofline_tail=rs["roofline_tail"],
                                              resident_single_shape_descriptors_per_s=rs["value"],
                                              resident_single_shape_backbone_ms_per_image=rs["backbone_ms_per_image"])
        else:
            extra["descriptors_per_s"] = {"value": None, "unit": "descriptors/s", "n_gpus": world,
                                          "error": err or "the extraction leg failed on another rank"}

    if rank == 0:
        qps = NQ * args.steps / elapsed
        line = {"metric": "queries/sec, exact full ranking (rOxford5k+1M-distractor shape, 2048-d fp32); mAP-medium alongside",
                "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32",
                "data": "synthetic" + (" (DRY RUN: all ranks on one GPU over gloo -- not a measurement)" if dryrun else ""),
                "config": {"workload": ("configs[2]: roxford5k+1M synthetic distractors, N=%d Q=%d D=%d, "
                                        "similarity + exact full ranking per step" % (n_total, NQ, DIM)) if world == 1 else
                                       ("configs[3]: the configs[2] database (N=%d Q=%d D=%d) row-sharded x%d, exchange of the per-shard partial "
                                        "scores over %s, query-split exact full ranking per step" % (n_total, NQ, DIM, world, "gloo (dry run)" if dryrun else "RCCL/xGMI")),
                           "db_rows_per_gpu": n_local, "parallelism": "db-row-shard x%d, query-split sort" % world,
                           "index_build_s": round(build_s, 4)}}
        line.update(extra)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
