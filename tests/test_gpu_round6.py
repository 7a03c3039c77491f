"""Round-6 GPU tests: the direct-store exchange (mdx_p2p_* / mdx_scores_p2p) with ranks in one process and as 2 / 8 rank
processes on this GPU, the 16 peer blocks of a G = 8 step through the segment sort, the float64 arbiter of bench.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import chain as OC
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ------------------------------------------------------------------------------------------------ direct-store exchange
@pytest.mark.parametrize("shape", [(70001, 23, 256, 3), (40000, 70, 128, 4), (3000, 5, 64, 2)])
def test_p2p_ranks_in_one_process_equal_the_oracle(shape):
    """`mdx_scores_p2p` + `mdx_p2p_close_step` through ops.P2P with all ranks in this process (connected by pointers, one
    stream per rank): after a step every rank holds ITS queries' rows of the WHOLE score matrix -- bit for bit the chain
    oracle's -- and their ranking is `OC.rank_full`'s.  Three steps with other queries: the two receive buffers alternate and
    the first step's view is overwritten by the third, not by the second.  Shapes: 128-row workgroups with the 4x4x1 leftover
    tile (23 queries), 64-row workgroups with one query tile per workgroup (70 queries x 10 000-row shards), tiny shards.  (At most
    four ranks here: ranks that live in ONE process are streams of one process, HIP multiplexes them onto four hardware queues, and a
    rank spinning for a peer's flag would hold the queue that peer's kernels wait in.  Rank PROCESSES have queues of their own: the
    8-rank cases below.)"""
    from mdir_amd import ops
    from mdir_amd.sharded import shard_bounds
    n, nq, d, G = shape
    vecs, qvecs, _ = O.synth_ranking_problem(n, nq, d, seed=n % 97)
    vecs[:, 7] = vecs[:, 3]
    vecs[:, n - 1] = vecs[:, 3]                                     # ties inside a shard and across shards
    rows = np.ascontiguousarray(vecs.T)
    peers = [ops.P2P(G, r, nq, n, DEV) for r in range(G)]
    for p in peers:
        p.connect_local(peers)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(G)]
    shards = []
    for r in range(G):
        lo, hi = shard_bounds(n, G, r)
        cut = lo + (hi - lo) // 3                                   # every rank holds its rows as two chunks: two launches per step
        shards.append([ops.DescriptorIndex(dev(rows[lo:cut]), "ND", lo), ops.DescriptorIndex(dev(rows[cut:hi]), "ND", cut)])
    rng = np.random.default_rng(5)
    first_views = None
    for step in range(3):
        q = qvecs if step == 0 else (qvecs + 0.3 * rng.standard_normal(qvecs.shape).astype(np.float32))
        qd = dev(q)
        torch.cuda.synchronize()
        mine = []
        for r in range(G):
            with torch.cuda.stream(streams[r]):
                for ix in shards[r]:
                    ix.scores_p2p(qd, peers[r], "DN")
                mine.append(peers[r].close_step())
        torch.cuda.synchronize()
        want = OC.scores_chain(vecs, q)                             # [nq, n]
        want_rk = OC.rank_full(want)
        for r in range(G):
            assert peers[r].late_peers() == 0
            qlo, qhi = peers[r].qlo, peers[r].qhi
            assert tuple(mine[r].shape) == (qhi - qlo, n)
            np.testing.assert_array_equal(mine[r].cpu().numpy(), want[qlo:qhi])
            if qhi > qlo:
                np.testing.assert_array_equal(ops.rank_full(mine[r]).cpu().numpy(), want_rk[qlo:qhi])
        if step == 0:
            first_views, first_want = mine, want
        if step == 1:                                               # the other buffer was written: step 0's views still hold step 0
            for r in range(G):
                np.testing.assert_array_equal(first_views[r].cpu().numpy(), first_want[peers[r].qlo:peers[r].qhi])
    for r in range(G):
        for ix in shards[r]:
            ix.close()
        peers[r].close()


def test_p2p_argument_checks():
    from mdir_amd import ops
    p = ops.P2P(1, 0, 5, 1000, DEV)
    ix = ops.DescriptorIndex(dev(np.zeros((1000, 64), np.float32)), "ND", 0)
    q = dev(np.zeros((5, 64), np.float32))
    with pytest.raises(ValueError, match="not connected"):
        ix.scores_p2p(q, p, "ND")
    p.connect([p.handle])
    ix.scores_p2p(q, p, "ND")
    with pytest.raises(ValueError, match="another number of queries"):
        ix.scores_p2p(dev(np.zeros((6, 64), np.float32)), p, "ND")
    half = ops.DescriptorIndex(dev(np.zeros((1000, 64), np.float32)), "ND", 0, storage="f16")
    with pytest.raises(ValueError, match="fp32 shard"):
        half.scores_p2p(q, p, "ND")
    with pytest.raises(ValueError):
        ops.P2P(2, 2, 5, 1000, DEV)
    with pytest.raises(ValueError):
        ops.P2P(2, 0, 129, 1000, DEV)
    assert tuple(p.close_step().shape) == (5, 1000) and p.late_peers() == 0
    p.close()


_RANKS_SCRIPT = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from mdir_amd.sharded import ShardedIndex, shard_bounds, query_bounds
from oracle import chain as OC
from oracle import oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n, nq, d = 70001, 23, 256
vecs, qvecs, _ = O.synth_ranking_problem(n, nq, d, seed=6)
vecs[:, 11] = vecs[:, 5]; vecs[:, n - 2] = vecs[:, 5]            # exact ties inside a shard and across shards
lo, hi = shard_bounds(n, world, rank)
sh = ShardedIndex(torch.from_numpy(np.ascontiguousarray(vecs[:, lo:hi])).cuda(), "DN", n)
assert sh._p2p_on
want_sc = OC.scores_chain(vecs, qvecs)                            # [nq, n]: the whole problem on the host
want_rk = OC.rank_full(want_sc)
for step in range(3):                                             # both receive buffers, and the first one again
    rk, sc, (qlo, qhi) = sh.rank_queries(torch.from_numpy(qvecs).cuda(), "DN")
    assert (qlo, qhi) == query_bounds(nq, world, rank)
    assert len(sc.blocks) == 1                                    # dense: no peer blocks
    assert np.array_equal(sc.dense().cpu().numpy(), want_sc[qlo:qhi]), "exchanged scores"
    assert np.array_equal(rk.cpu().numpy(), want_rk[qlo:qhi]), "global ranking ids"
assert sh._p2p.late_peers() == 0
dist.barrier()
sh._p2p.close()
dist.destroy_process_group()
print("P2P-RANK-OK", rank, flush=True)
"""


@pytest.mark.parametrize("world,chunks", [(2, "1"), (2, "2"), (8, "1")])
def test_rank_processes_on_one_gpu_with_the_direct_store_exchange(world, chunks, tmp_path):
    """`MDIR_AMD_COMM=p2p` with 2 and 8 rank PROCESSES sharing this GPU (same-device hipIpc: the functional form of the
    exchange; xGMI has never carried it): every rank's `ShardedIndex.rank_queries` -- similarity kernels storing into the owners'
    buffers, one flag per peer, dense ranking -- equals `OC.rank_full` of the whole problem to the last id, three steps in a row."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "p2p_ranks.py"
    script.write_text(_RANKS_SCRIPT % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", MDIR_AMD_COMM="p2p", MDIR_AMD_EXCHANGE_CHUNKS=chunks)
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                           "127.0.0.1", "--master-port", str(port), str(script)], env=env, text=True, capture_output=True, timeout=1200)
    if proc.returncode != 0:                                       # (the whole output, for the log of a GPU box)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "p2p_ranks_failure_%d_%s.log" % (world, chunks)), "w") as f:
            f.write(proc.stdout + "\n=== stderr ===\n" + proc.stderr)
    assert proc.returncode == 0 and proc.stdout.count("P2P-RANK-OK") == world, (proc.stdout[-2000:], proc.stderr[-4000:])


# ------------------------------------------------------------------------------------------------ a G = 8 step's peer blocks
def test_sixteen_peer_blocks_of_a_g8_step_rank_like_the_dense_matrix():
    """One rank's ranking at G = 8: 9 queries x 1 004 993 rows delivered as 8 peers x 2 chunks (65 536 + the rest: `chunk_bounds`) --
    every sixteenth sort tile ends in the next block (the two-block path of round 6).  Equal to `mdx_rank_full` of the
    concatenated matrix and, on three rows, to the C oracle."""
    from mdir_amd import ops
    from mdir_amd.sharded import chunk_bounds, shard_bounds
    n, G = 1004993, 8
    g = torch.Generator(device=DEV)
    g.manual_seed(16)
    full = torch.randn((9, n), generator=g, device=DEV) * 0.022
    full[:, 65536] = full[:, 65535]                                 # ties across a block border
    blocks = []
    for r in range(G):
        for a, b in chunk_bounds(*shard_bounds(n, G, r), 2):
            blocks.append(full[:, a:b].contiguous())
    assert len(blocks) == 16 and blocks[0].shape[1] == 65536
    got = ops.rank_full_segments(blocks)
    assert torch.equal(got, ops.rank_full(full))
    host = full[[0, 4, 8]].cpu().numpy()
    np.testing.assert_array_equal(got[[0, 4, 8]].cpu().numpy(), OC.rank_full(host))


# ------------------------------------------------------------------------------------------------ float64 arbiter
def test_float64_arbiter_at_70_x_200000():
    """bench.py's `cpu_path_parity.f64_arbiter` on the bench's own synthetic workload at 70 x 200 000 x 2048: where the GPU chain
    and the host's BLAS + argsort disagree, the float64 evaluation of the same statement (cirscore.py:69-70) decides; the GPU
    chain is never further from the float64 scores than the summation-order bound 2e-6, and two rows it orders differently
    from float64 are never further apart than that bound in float64.  The device's float64 values equal numpy's on the host."""
    sys.path.insert(0, ROOT)
    import bench
    from mdir_amd import ops
    n = 200_000
    device = torch.device(DEV)
    rows = bench.gen_rows(0, n, device)
    qvecs, qid = bench.gen_queries(n, device)
    gnd = bench.synth_gnd(min(bench.N_ROXFORD, n))
    bench.plant_positives(rows, 0, n, gnd, qid, device)
    ix = ops.DescriptorIndex(rows, "ND")
    sc = ix.scores(qvecs, "DN")
    rk = ops.rank_full(sc)
    vecs_host = rows.t().contiguous().cpu().numpy()
    sc_cpu, rk_cpu, _, _ = bench.cpu_baseline(vecs_host, qvecs.cpu().numpy(), reps=1)
    arb = bench.f64_arbiter(rows, qvecs, sc, rk, rk_cpu, sc_cpu, gnd, vecs_host)
    tol = bench.SUM_ORDER_TOL
    assert arb["gpu_max_abs_score_error_vs_f64"] <= tol and arb["cpu_max_abs_score_error_vs_f64"] <= tol
    assert arb["gpu_max_f64_gap_between_misordered_rows"] <= tol
    # the GPU chain is not further from the float64 order than BLAS by more than the bound (both are inside it)
    assert arb["gpu_max_f64_gap_between_misordered_rows"] <= arb["cpu_max_f64_gap_between_misordered_rows"] + tol
    nd = arb["slots_where_gpu_and_cpu_differ"]
    assert arb["gpu_order_agrees_with_f64"] + arb["cpu_order_agrees_with_f64"] + arb["neither_agrees_with_f64"] >= nd
    assert arb["gpu_order_agrees_with_f64"] <= nd and arb["cpu_order_agrees_with_f64"] <= nd
    assert arb["of_slots"] == 70 * n
    lab = arb["labelled_rows"]
    assert lab["of"] == 20 * 70 and lab["gpu_position_equals_f64"] <= lab["of"]
    assert 0.0 < arb["map_medium_f64_order"] <= 1.0
    if "host_f64_crosscheck" in arb:
        assert arb["host_f64_crosscheck"]["max_abs_diff_device_f64_vs_numpy_f64"] <= 1e-12
    # the three orders agree on nearly everything: a regression that reorders rows wholesale cannot hide here
    assert arb["top100_slots_equal_to_f64"]["gpu"] >= 0.98 * 7000 and arb["whole_ranking_slots_equal_to_f64"]["gpu"] >= 0.3 * 70 * n
    ix.close()


# ------------------------------------------------------------------------------------------------ preflight + run-time choice
def test_preflight_forms_as_fresh_rank_processes_on_one_gpu():
    """tools/preflight_ranks.py as bench.py starts it (fresh children; here 2 ranks on this GPU over gloo): every exchange form
    passes its 10 000-row end-to-end check, the verdict names what ran, and a form that cannot work falls through to the next."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import preflight_ranks
    pre = preflight_ranks.run(2, "p2p", dryrun=True)
    assert pre["form"] == "p2p" and pre["tried"][0]["ok"] and pre["tried"][0]["nranks_seen"] == 2
    assert pre["tried"][0]["exchange_used"] == "direct_store" and "link_types" in pre["tried"][0]
    pre = preflight_ranks.run(2, "torch", dryrun=True)
    assert pre["form"] == "torch" and pre["tried"][0]["exchange_used"] == "all_to_all"
    assert preflight_ranks.fallbacks("mdx") == ["mdx", "torch", "allgather"]
    # a form that fails (here: the direct-store exchange without dmabuf IPC) is reported with its reason and the next one is taken
    old = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "1"
    try:
        pre = preflight_ranks.run(2, "p2p", dryrun=True)
    finally:
        if old is None:
            del os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]
        else:
            os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = old
    assert pre["form"] in ("p2p", "mdx", "torch", "allgather")
    if pre["form"] != "p2p":
        assert not pre["tried"][0]["ok"] and pre["tried"][0]["reason"]


def test_bench_two_ranks_with_preflight_and_exchange_selection():
    """`bench.py --gpus 2` as a dry run WITH its preflight (the default): the line records which form passed, the direct-store
    probe, and the run-time A/B of the two forms at the run's size (`exchange_selection`: both timings, bit-identical rankings,
    the choice); mAP as in the single-process line."""
    import json
    env = dict(os.environ, MDIR_AMD_DRYRUN_ONE_GPU="1")
    args = ["--gpus", "2", "--rows", "150000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--extract-images", "0"]
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, text=True, capture_output=True, timeout=1500)
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-4000:])
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["preflight"]["form_that_passed"] == "torch" and line["preflight"]["tried"][0]["ok"]
    assert line["preflight"]["direct_store_probe"]["ok"] is True
    sel = line["exchange_selection"]
    assert sel["direct_store_verified_equal"] is True and sel["chosen"] in ("direct_store", "collective")
    assert sel["collective_ms"] > 0 and sel["direct_store_ms"] > 0
    assert line["phases_ms_per_rank"]["exchange"] == ("direct_store" if sel["chosen"] == "direct_store" else "all_to_all")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "150000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--extract-images", "0"], text=True, capture_output=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    assert json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])["map_medium"] == line["map_medium"]

