#!/usr/bin/env python3
"""Headline benchmark: queries/sec of the ranking hot path on the rOxford5k +
1M-distractor shaped workload (BASELINE.json configs[2]; configs[3] for --gpus N).

A "step" = one pass of the hot path over one batch of 70 queries against the
resident database: similarity (mdx_scores, fp32 MFMA) + exact full ranking
(mdx_rank_full), i.e. the reference's `np.dot(vecs.T, qvecs)` + `np.argsort(-scores,
axis=0)` (mdir/components/optim/score/cirscore.py:69-70).  Inputs are resident in
HBM when the timed region starts.  With --gpus N the 1M database is row-sharded
(strong scaling); see mdir_amd/sharded.py for the exchange.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows ROWS] [--no-cpu-baseline] [--secondary]
                    [--extract-images M] [--profile] [--comm torch|mdx|p2p]

The default run = the headline loop + `cpu_baseline` (with the float64 arbiter of `cpu_path_parity`) + the descriptors/s leg
(with its own `cpu_baseline` and trunk `roofline`).  `--secondary` appends the side legs of tools/bench_secondary.py.
`--profile` = the headline loop alone (no CPU baseline, side legs or extraction): the form tools/profile_round.sh runs under
rocprofv3, so that the per-kernel averages of the committed profile add up to the step.

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
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch
import torch.distributed as dist

N_ROXFORD, N_DISTRACTORS = 4993, 1_000_000
NQ, DIM = 70, 2048
GEN_BLOCK = 4096
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0
from bench_parity import SUM_ORDER_TOL, cpu_path_parity, f64_arbiter          # noqa: E402,F401  (tools/: the parity half of the cpu_baseline leg)
from bench_ranks import launch_ranks, preflight, ranks_report, requested_form, select_exchange      # noqa: E402,F401  (tools/: the N > 1 half)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", dest="n", type=int, default=N_ROXFORD + N_DISTRACTORS, help="database rows (default 1 004 993)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", action="store_true",
                    help="also run the side legs of tools/bench_secondary.py (configs[1] / configs[4] shapes, split-precision modes, top-100, "
                         "one evaluation, whitening learning, CLAHE, the sort-free route) and append them to the line")
    ap.add_argument("--no-secondary", action="store_true", help="(accepted for older command lines: the side legs are off by default)")
    ap.add_argument("--extract-images", type=int, default=40,
                    help="images PER SIZE (16 sizes) of the (untimed) descriptors/sec leg: ResNet101-GeM, 3 scales + whitening; 0 = skip")
    ap.add_argument("--no-pipelined", action="store_true", help="(accepted for older command lines; no effect)")
    ap.add_argument("--profile", action="store_true", help="the headline loop only: --no-cpu-baseline --extract-images 0")
    ap.add_argument("--no-preflight", action="store_true", help="N > 1: skip tools/preflight_ranks.py (fresh child processes that check the "
                    "exchange form on a small problem before the heavy run) and the run-time choice between the exchange forms")
    ap.add_argument("--comm", choices=("torch", "mdx", "p2p"), default=None,
                    help="N > 1: the exchange of partial scores through torch.distributed (default), through the C-ABI communicator "
                         "(mdx_comm_* / mdx_exchange_scores over RCCL; same as MDIR_AMD_COMM=mdx), or with no collective at all: the "
                         "similarity kernel stores into the owners' receive buffers (mdx_scores_p2p, hipIpc + xGMI stores; MDIR_AMD_COMM=p2p)")
    args = ap.parse_args()
    if args.profile:
        args.no_cpu_baseline, args.secondary = True, False
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
    pre = json.loads(os.environ["MDIR_AMD_PREFLIGHT"]) if os.environ.get("MDIR_AMD_PREFLIGHT") else None
    if world > 1 and pre is None and rank == 0 and not args.no_preflight:
        # launched as ranks (the driver's torch.distributed.run): rank 0 runs the preflight's fresh children BEFORE it touches the
        # GPU; the other ranks wait for it in init_process_group below and receive the verdict there
        pre = preflight(world)
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
        # the ranks that wait while rank 0 times the CPU reference do so on a gloo side group with its own generous limit: the
        # baseline must not be able to cost the measured line through the 900 s limit of the data-path group
        wait_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=2))
        if not args.no_preflight:
            box = [pre]
            dist.broadcast_object_list(box, src=0)
            pre = box[0]
            # no form passed: more likely the preflight itself could not run here (children refused, a time limit) than all four
            # forms being broken -- the measured line must not depend on it: keep the requested form (ShardedIndex still probes the
            # all-to-all in this process and falls back to the all-gather on every rank together), and say so in the line
            if pre is not None and pre["form"] is not None:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import preflight_ranks
                os.environ.update(preflight_ranks.form_env(pre["form"]))       # the form that passed (the requested one or a fall-back)
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
    selection = None
    if world > 1 and pre is not None and pre.get("p2p_probe", {}).get("ok") and not sharded._p2p_on and sharded.storage == "f32":
        selection = select_exchange(sharded, step, keep, dryrun, device, qvecs)
    def timed_region():
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
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dryrun else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    elapsed = timed_region()
    if world > 1 and selection is not None and selection["chosen"] == "direct_store":
        # the form chosen at run time must also have held THROUGH the timed steps: every rank's last ranking verified on its
        # device, no flag wait given up -- else the collective form is measured instead (the line says so), never a lost line
        good = 1
        try:
            rk_c, sc_c, (ql_c, qh_c) = keep["rk"], keep["sc"], keep["q"]
            if qh_c > ql_c:
                p_ok, o_ok = verify_ranking(sc_c.dense(), rk_c)
                good = int(p_ok and o_ok)
            if sharded._p2p is not None and sharded._p2p.late_peers() != 0:
                good = 0
        except Exception:           # noqa: BLE001
            good = 0
        flag = torch.tensor([good], dtype=torch.int32, device="cpu" if dryrun else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            sharded.use_direct_store(False)
            selection["chosen"] = "collective"
            selection["reason"] = "the direct-store form did not verify after the timed steps; the timed region was repeated with the collective form"
            for _ in range(args.warmup):
                step()
            elapsed = timed_region()

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
        if prof.get("scores_kernel_trace_avg_us"):
            # the second reading of the same kernel: its average duration in the committed rocprofv3 kernel trace (same command
            # under the profiler, same sources by hash).  HIP events in this run and the trace of that run sit side by side.
            us = prof["scores_kernel_trace_avg_us"]
            roofline["rocprof_kernel_avg_us"] = us
            roofline["frac_rocprof"] = round(flops / (us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
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
            "frac_rocprof": (round(rank_algo / (prof["ranking_trace_sum_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)
                             if prof.get("ranking_trace_sum_us") and rank_traffic else None),
            "traffic_over_algorithmic": round(rank_traffic / rank_algo, 2) if rank_traffic else None,
            "hbm_GBps_at_real_traffic": round(rank_traffic / (rank_ms * 1e-3) / 1e9, 1) if rank_traffic else None,
            "what": "12 launches per ranking; the 4-pass form moves ~4.8x the bytes of an ideal one-pass argsort and streams them at the "
                    "rate HBM gives a read+write mix (DESIGN section 4): only fewer passes would help, and the MSD / one-sweep forms measured slower"}
    else:
        import types
        extra.update(ranks_report(types.SimpleNamespace(keep=keep, dryrun=dryrun, device=device, qid=qid, gnd=gnd, sharded=sharded, qvecs=qvecs,
                                                         world=world, n_total=n_total, ev=ev, args=args, pre=pre, selection=selection)))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            vecs_host = rows.t().contiguous().cpu().numpy()       # reference layout [D,N]
            sc_cpu, rk_cpu, t_dot, t_sort = cpu_baseline(vecs_host, qvecs.cpu().numpy())
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
            # parity with the reference CPU path at full size + the float64 arbiter (tools/bench_parity.py)
            cpu_path_parity(rows, qvecs, sc, rk, sc_cpu, rk_cpu, gnd, vecs_host, extra, n_total, NQ)
            del vecs_host, rk_cpu, sc_cpu
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
        dist.barrier(group=wait_group)

    if rank == 0 and world == 1 and args.secondary:
        # BASELINE.json's other single-GPU configurations and the side legs of earlier rounds (tools/bench_secondary.py): NOT part
        # of the default run since round 6 -- they cannot cost the measured line, and they no longer cost the driver's minutes
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_secondary
        try:
            bench_secondary.sort_free_leg(args, sharded, qvecs, sc, gnd, device, extra, NQ)
            extra["secondary_configs"] = bench_secondary.secondary_configs(args, ops, sharded, rows, qvecs, sc, rk, ws, gnd, device, extra,
                                                                           n_total, NQ, DIM)
        except Exception as exc:
            extra["secondary_configs"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if args.extract_images > 0:
        # second half of BASELINE.json's metric: descriptors/sec (every rank extracts its own images).  The number is
        # taken on an image LIST -- 16 JPEG sizes through the real loader -- in steady state; what a new size costs
        # and the resident single-shape figure of round 1 are reported beside it.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import types
        from bench_extract import cpu_reference_loop, measure, measure_list
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
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                # the reference-style extraction loop on this box's host cores (bounded sample; baseline only)
                try:
                    extra["descriptors_per_s"]["cpu_baseline"] = cpu_reference_loop("resnet101", images=8)
                except Exception as exc:
                    extra["descriptors_per_s"]["cpu_baseline"] = {"value": None, "unit": "descriptors/s", "error": "%s: %s" % (type(exc).__name__, exc)}
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
