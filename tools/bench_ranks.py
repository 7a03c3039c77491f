"""The N > 1 half of bench.py: starting the rank processes, the preflight of the exchange forms (tools/preflight_ranks.py),
the run-time A/B of the collective and the direct-store form, and the per-rank part of the line (`ranks_report`)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def requested_form():
    """The exchange form the command line / environment asks for, in tools/preflight_ranks.py's vocabulary."""
    comm = os.environ.get("MDIR_AMD_COMM") or ""
    if comm in ("p2p", "mdx"):
        return comm
    return "allgather" if os.environ.get("MDIR_AMD_EXCHANGE") == "allgather" else "torch"


def preflight(n):
    """tools/preflight_ranks.py with FRESH child processes (called by a process that has not touched the GPU): the requested
    exchange form and its fall-backs on a 10 000-row problem, every rank's rows verified on the device; then -- unless the direct-store
    form was the one requested, or MDIR_AMD_COMM_AUTO=0 -- ONE probe of the direct-store form, whose verdict decides whether the
    heavy run may time it against the collective (`exchange_selection`)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    first = requested_form()
    try:
        import preflight_ranks
        pre = preflight_ranks.run(n, first)
        if (pre["form"] is not None and pre["form"] != "p2p" and first == "torch" and os.environ.get("MDIR_AMD_COMM_AUTO", "1") != "0"
                and pre["seconds"] < 200):
            pre["p2p_probe"] = preflight_ranks.run(n, "p2p", only=True, budget_s=160)["tried"][0]
    except Exception as exc:          # noqa: BLE001 -- a preflight that cannot run must not cost the measured line
        pre = {"form": None, "tried": [{"form": first, "ok": False, "reason": "the preflight itself failed: %s: %s" % (type(exc).__name__, exc)}],
               "seconds": None}
    return pre


def select_exchange(sharded, step, keep, dryrun, device, qvecs, reps=3):
    """Run time A/B of the two exchange forms on THIS node at the full size, outside the timed region: `reps` steps of the
    collective form, then of the direct-store form (which the preflight has just verified on a small problem with fresh
    processes).  The direct-store form is taken only if every rank's ranking is bit-identical to the collective form's, no
    flag wait gave up, and the slowest rank's step is faster.  Any failure leaves the collective form in place."""
    def timed():
        step()                                              # (first step of a form: buffers, peer mappings)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], dtype=torch.float64, device="cpu" if dryrun else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    out = {"collective_ms": round(timed(), 4)}
    rk_a = keep["rk"].clone()
    ok, why = 1, ""
    # set-up first, with the ranks in lockstep (every collective of it is reached by every rank, whatever fails locally): after it
    # a direct-store step contains no collective, so a rank that fails INSIDE the trial cannot strand the others in one
    nq = int(qvecs.shape[1])
    if not sharded.prepare_direct_store(nq):
        sharded.use_direct_store(False)
        out.update({"direct_store_verified_equal": False, "chosen": "collective",
                    "reason": "the direct-store exchange could not be set up on every rank"})
        return out
    sharded.use_direct_store(True)
    trouble, real_step = [], step

    def guarded_step():
        if not trouble:
            try:
                real_step()
            except Exception as exc:          # noqa: BLE001 -- this rank sits the rest of the trial out, the barriers below still match
                trouble.append("%s: %s" % (type(exc).__name__, exc))
    step = guarded_step                       # (what timed() calls)
    try:
        out["direct_store_ms"] = round(timed(), 4)
    finally:
        step = real_step
    if trouble:
        ok, why = 0, trouble[0]
    elif not torch.equal(keep["rk"], rk_a):
        ok, why = 0, "the direct-store ranking differs from the collective form's"
    elif sharded._p2p is not None and sharded._p2p.late_peers() != 0:
        ok, why = 0, "a peer's flag did not arrive"
    flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if dryrun else device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = int(flag.item())
    take = bool(ok and out.get("direct_store_ms", 1e9) < out["collective_ms"])
    sharded.use_direct_store(take)
    out.update({"direct_store_verified_equal": bool(ok), "chosen": "direct_store" if take else "collective",
                "what": "per-step wall time of the slowest rank over %d steps of each form at the full size, before the timed region" % reps})
    if why:
        out["reason"] = why
    return out


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
    if "--no-preflight" not in sys.argv and "MDIR_AMD_PREFLIGHT" not in env:
        # this process never touches the GPU: the preflight's fresh children run here, and the heavy run's ranks are told the verdict
        pre = preflight(n)
        if pre["form"] is None:         # (the heavy run then keeps the requested form, as if there had been no preflight: see main)
            print("bench.py --gpus %d: no exchange form passed the preflight: %s" % (n, json.dumps(pre["tried"])), file=sys.stderr)
        env["MDIR_AMD_PREFLIGHT"] = json.dumps(pre)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def ranks_report(c):
    """Everything of the N > 1 line that is not the headline figure: verification of every rank's rows on its device, mAP
    without a ranking, per-rank rooflines and phases, preflight / selection records.  `c`: the namespace bench.main fills."""
    from bench import (DIM, N_DISTRACTORS, N_ROXFORD, NQ, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, committed_traffic, spread, verify_ranking)
    keep, dryrun, device, qid, gnd, sharded, qvecs = c.keep, c.dryrun, c.device, c.qid, c.gnd, c.sharded, c.qvecs
    world, n_total, ev, args, pre, selection = c.world, c.n_total, c.ev, c.args, c.pre, c.selection
    extra = {}
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
    if pre is not None:
        extra["preflight"] = {"form_that_passed": pre["form"], "seconds": pre.get("seconds"),
                              "tried": [{k: v for k, v in t.items() if k in ("form", "ok", "reason", "nranks_seen", "link_types", "exchange_used", "through", "seconds_in_ranks")}
                                        for t in pre["tried"]],
                              "direct_store_probe": {k: v for k, v in pre.get("p2p_probe", {}).items() if k in ("ok", "reason", "nranks_seen", "seconds_in_ranks")} or None}
    if selection is not None:
        extra["exchange_selection"] = selection
    extra["comm"] = ("p2p (C ABI: mdx_scores_p2p, direct stores into the owners' buffers + one flag per peer)" if getattr(sharded, "_p2p_on", False)
                     else "mdx (C ABI: mdx_exchange_scores over RCCL)" if getattr(sharded, "_comm", None) is not None else "torch.distributed")
    if getattr(sharded, "_p2p", None) is not None:
        extra["p2p_late_peers"] = sharded._p2p.late_peers()
    extra["phases_ms_per_rank"] = {"scores": [round(float(x), 4) for x in table[:, 0]],
                                   "exchange_exposed": [round(float(x), 4) for x in table[:, 1]],
                                   "sort": [round(float(x), 4) for x in table[:, 2]],
                                   "exchange": "direct_store" if getattr(sharded, "_p2p_on", False) else ("all_to_all" if sharded._use_a2a else "all_gather"),
                                   "chunks": sharded.chunks,
                                   "what": "last timed step; exchange_exposed = compute-stream wait for transfers after the last "
                                           "similarity kernel (+ re-block copy)"}
    return extra
