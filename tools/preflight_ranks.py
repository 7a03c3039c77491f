#!/usr/bin/env python3
"""A one-minute preflight for the first real N > 1 run.

RCCL over xGMI has never carried more than one rank of this code: the first 8-GPU run is also the first integration test, and
a trivial failure there (an exchange form the node does not support, a peer mapping that fails) would cost the measured line.
This script is that integration test in small: run as N rank processes it

  * initialises the process group (RCCL; gloo under MDIR_AMD_DRYRUN_ONE_GPU=1, all ranks on one GPU) and counts the ranks it sees,
  * (rank 0) records the link types `rocm-smi --showtopo` reports,
  * runs ONE exchange form end to end on a 10 000-row database with the real kernels -- `ShardedIndex.rank_queries`: shard
    similarity, exchange, query-split ranking -- and verifies every rank's rows on the device (a permutation of the global ids,
    non-increasing scores, ascending ids inside ties) and the head of each row against the scores it was ranked from,
  * prints one JSON line (rank 0) and exits 0 / 3.

Forms: `p2p` (mdx_scores_p2p: direct stores into the owners' buffers), `mdx` (mdx_exchange_scores over RCCL), `torch`
(torch.distributed all_to_all_single), `allgather` (the literal all-gather of partial scores).

    python -m torch.distributed.run --nproc-per-node N tools/preflight_ranks.py --form torch      (what `run()` below starts)

`run(n, forms)` -- used by bench.py BEFORE its heavy run -- starts FRESH children for each form in turn until one passes (never
a re-exec of a process that has touched the GPU) and returns the verdicts; bench.py then sets the exchange form accordingly and
records it in its line (`preflight`).  A form that hangs is ended by the children's own limits (60 s process-group timeout, 20 s
flag waits) and by the parent's `timeout`.
"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMS = ("p2p", "mdx", "torch", "allgather")
N_ROWS, NQ, DIM = 10_000, 70, 256


def form_env(form):
    """The environment that makes mdir_amd.sharded.ShardedIndex use `form`."""
    env = {"MDIR_AMD_COMM": "", "MDIR_AMD_EXCHANGE": ""}
    if form == "p2p":
        env["MDIR_AMD_COMM"] = "p2p"
    elif form == "mdx":
        env["MDIR_AMD_COMM"] = "mdx"
    elif form == "allgather":
        env["MDIR_AMD_EXCHANGE"] = "allgather"
    elif form != "torch":
        raise ValueError("form %r (one of %s)" % (form, ", ".join(FORMS)))
    return env


def fallbacks(first):
    """The order in which forms are tried, starting from the requested one."""
    order = {"p2p": ["p2p", "mdx", "torch", "allgather"], "mdx": ["mdx", "torch", "allgather"],
             "torch": ["torch", "allgather"], "allgather": ["allgather"]}
    return order[first]


def run(n, first="torch", dryrun=None, timeout_s=150, python=sys.executable, only=False, budget_s=320):
    """Preflight with FRESH child processes: every form of `fallbacks(first)` in turn until one passes.  Returns
    ``{"form": <the form that passed or None>, "tried": [verdict per form], "seconds": ...}``.  The caller must not have touched the
    GPU in a way that forbids starting children (this function only starts processes; it never execs).  ``only``: just `first`,
    no fall-backs (a probe of one form)."""
    dryrun = os.environ.get("MDIR_AMD_DRYRUN_ONE_GPU") == "1" if dryrun is None else dryrun
    t0 = time.perf_counter()
    tried, chosen = [], None
    for form in ([first] if only else fallbacks(first)):
        left = budget_s - (time.perf_counter() - t0)
        if left < 30:                                   # the preflight must stay a preflight: what is left of its budget bounds every form
            tried.append({"form": form, "ok": False, "reason": "not tried: the preflight's %d s were used up" % budget_s})
            continue
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("OMP_NUM_THREADS", "2")
        env.update(form_env(form))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK",
                  "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
                  "TORCHELASTIC_USE_AGENT_STORE", "MDIR_AMD_EXCHANGE_CHUNKS"):
            env.pop(k, None)                          # the children are a job of their own, not ranks of the caller's
        if dryrun:
            env["MDIR_AMD_DRYRUN_ONE_GPU"] = "1"
        cmd = [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), "--form", form]
        verdict = {"form": form, "ok": False}
        # the launcher and its rank processes are a process group of their own: a form that hangs is ended AS A WHOLE (killing only
        # the launcher would orphan rank processes that keep their GPUs busy while the heavy run starts)
        proc = subprocess.Popen(cmd, env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            out, err = proc.communicate(timeout=min(timeout_s, left))
            lines = [ln for ln in out.splitlines() if ln.startswith("{")]
            if lines:
                verdict.update(json.loads(lines[-1]))
            verdict["ok"] = bool(proc.returncode == 0 and verdict.get("ok"))
            if not verdict["ok"]:
                verdict.setdefault("reason", "exit code %d: %s" % (proc.returncode, (err or out)[-600:]))
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)          # exactly the group this call started
            except (ProcessLookupError, PermissionError):
                pass
            proc.communicate()
            verdict["reason"] = "no answer within %d s (launcher and rank processes ended)" % min(timeout_s, left)
        tried.append(verdict)
        if verdict["ok"]:
            chosen = form
            break
    return {"form": chosen, "tried": tried, "seconds": round(time.perf_counter() - t0, 1)}


def topo_links():
    """Link types between GPU pairs as `rocm-smi --showtopo` prints them (XGMI / PCIE), or the reason there are none."""
    try:
        out = subprocess.run(["rocm-smi", "--showtopo"], text=True, capture_output=True, timeout=30).stdout
    except Exception as exc:                # noqa: BLE001 -- a report, not a requirement
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    kinds = {}
    seen = False
    for line in out.splitlines():
        if "Link Type between two GPUs" in line:
            seen = True
            continue
        if seen and line.strip().startswith("GPU") and ("XGMI" in line or "PCIE" in line or "0" in line):
            for tok in line.split()[1:]:
                if tok in ("XGMI", "PCIE"):
                    kinds[tok] = kinds.get(tok, 0) + 1
        elif seen and line.startswith("="):
            if kinds:
                break
    return kinds or {"note": "no link table in the output (one GPU?)"}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--form", choices=FORMS, default="torch")
    args = ap.parse_args()
    os.environ.update(form_env(args.form))
    sys.path.insert(0, ROOT)
    import datetime

    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dryrun = os.environ.get("MDIR_AMD_DRYRUN_ONE_GPU") == "1"
    if dryrun:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    limit = datetime.timedelta(seconds=60)
    if world > 1:
        if dryrun:
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=device, timeout=limit)
    from mdir_amd.sharded import ShardedIndex, query_bounds, shard_bounds
    out = {"form": args.form, "world": world, "ok": False, "backend": "gloo (dry run on one GPU)" if dryrun else "nccl"}
    t0 = time.perf_counter()
    seen = torch.ones(1, device="cpu" if dryrun else device)
    if world > 1:
        dist.all_reduce(seen)
    out["nranks_seen"] = int(seen.item())
    if rank == 0:
        out["link_types"] = topo_links()
    # the problem: block-seeded unit rows (any rank can regenerate any row), queries = noisy copies of 70 rows
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    full = torch.randn((N_ROWS, DIM), generator=g, device=device)
    full /= full.norm(dim=1, keepdim=True)
    full[17] = full[9000]                                   # a tie across shards
    qrows = torch.arange(NQ, device=device) * (N_ROWS // NQ)
    q = full[qrows] + 0.05 * torch.randn((NQ, DIM), generator=g, device=device)
    q /= q.norm(dim=1, keepdim=True)
    lo, hi = shard_bounds(N_ROWS, world, rank)
    sh = ShardedIndex(full[lo:hi].contiguous(), "ND", N_ROWS)
    ok = 1
    why = ""
    try:
        for _ in range(3):                                  # three steps: both receive buffers of the direct-store form, and the first again
            rk, sc, (qlo, qhi) = sh.rank_queries(q.contiguous(), "ND")
        assert (qlo, qhi) == query_bounds(NQ, world, rank)
        s = sc.dense()
        if qhi > qlo:
            want = q[qlo:qhi] @ full.t()                    # (any fp32 order: compared within a tolerance, the ORDER is checked exactly below)
            assert float((s - want).abs().max()) <= 1e-5, "exchanged scores differ from the product"
            srt = torch.gather(s, 1, rk)
            assert bool((srt[:, :-1] >= srt[:, 1:]).all()), "scores do not descend along a ranking"
            assert bool(((srt[:, :-1] != srt[:, 1:]) | (rk[:, :-1] < rk[:, 1:])).all()), "ids do not ascend inside a tie"
            assert bool((torch.sort(rk, dim=1).values == torch.arange(N_ROWS, device=device)).all()), "a ranking is not a permutation"
            assert bool((rk[:, 0] == qrows[qlo:qhi]).all()), "a query does not retrieve its source row first"
        if getattr(sh, "_p2p", None) is not None:
            assert sh._p2p.late_peers() == 0, "a peer's flag never arrived"
    except Exception as exc:                # noqa: BLE001 -- the verdict of this form
        ok, why = 0, "%s: %s" % (type(exc).__name__, exc)
    flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if dryrun else device)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    out["ok"] = bool(int(flag.item()))
    out["exchange_used"] = "direct_store" if getattr(sh, "_p2p_on", False) else ("all_to_all" if sh._use_a2a else "all_gather")
    out["through"] = "mdx_scores_p2p" if getattr(sh, "_p2p_on", False) else ("mdx_comm (C ABI over RCCL)" if sh._comm is not None else "torch.distributed")
    whys = [why]
    if world > 1:
        whys = [None] * world
        dist.all_gather_object(whys, why)
    if any(whys):
        out["reason"] = "; ".join("rank %d: %s" % (r, w) for r, w in enumerate(whys) if w)
    out["seconds_in_ranks"] = round(time.perf_counter() - t0, 2)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if getattr(sh, "_p2p", None) is not None:
            sh._p2p.close()
        dist.destroy_process_group()
    sys.exit(0 if out["ok"] else 3)


if __name__ == "__main__":
    main()
