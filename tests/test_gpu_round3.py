"""Round-3 additions on an MI355X: the f64 whitening-learning kernels, the RCCL exchange behind the C ABI, the
self-launching `bench.py --gpus N`, both wave-rank forms of the sort, and the ranking under hipGraph capture."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import chain as OC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ---------------------------------------------------------------- f3: f64 Gram / projection

@pytest.mark.parametrize("d,n", [(64, 16), (24, 600), (130, 77), (257, 1000), (512, 3), (2048, 640)])
def test_gram_f64_vs_numpy(d, n):
    """mdx_gram_f64 = np.dot(A, A.T) of whiten.py:22,42,46 in float64 (summation order differs from BLAS: rtol 1e-12 of
    the row norms), exactly symmetric, with the centring of whiten.py:21 fused."""
    from mdir_amd import ops
    rng = np.random.default_rng(d + n)
    A = rng.standard_normal((d, n))
    scale = np.sqrt(np.outer((A * A).sum(1), (A * A).sum(1)))
    got = ops.gram_f64(dev(A)).cpu().numpy()
    assert np.max(np.abs(got - A @ A.T) / scale) < 1e-13
    np.testing.assert_array_equal(got, got.T)
    m = A.mean(axis=1)
    Ac = A - m[:, None]
    got_c = ops.gram_f64(dev(A), dev(m)).cpu().numpy()
    scale_c = np.sqrt(np.outer((Ac * Ac).sum(1), (Ac * Ac).sum(1))) + 1e-300
    assert np.max(np.abs(got_c - Ac @ Ac.T) / scale_c) < 1e-13


@pytest.mark.parametrize("dout,d,n", [(24, 24, 600), (64, 64, 64), (100, 130, 77), (2048, 2048, 333), (5, 300, 1000)])
def test_project_f64_vs_numpy(dout, d, n):
    """mdx_project_f64 = np.dot(P, X - m) of whiten.py:45."""
    from mdir_amd import ops
    rng = np.random.default_rng(dout + d + n)
    P, X, m = rng.standard_normal((dout, d)), rng.standard_normal((d, n)), rng.standard_normal(d)
    want = P @ (X - m[:, None])
    got = ops.project_f64(dev(P), dev(X), dev(m)).cpu().numpy()
    bound = np.sqrt((P * P).sum(1))[:, None] * np.sqrt(((X - m[:, None]) ** 2).sum(0))[None, :]
    assert np.max(np.abs(got - want) / bound) < 1e-13
    got0 = ops.project_f64(dev(P), dev(X)).cpu().numpy()
    assert np.max(np.abs(got0 - P @ X) / (np.sqrt((P * P).sum(1))[:, None] * np.sqrt((X * X).sum(0))[None, :])) < 1e-13


def test_whitening_learning_golden_through_native_kernels(golden):
    """Golden G12 (the reference's whitenlearn / pcawhitenlearn on float64 descriptors) through mdx_gram_f64 /
    mdx_project_f64; rows of P are eigenvector-derived, hence compared up to sign."""
    from mdir_amd.whiten import pcawhitenlearn, whitenlearn
    g = golden("g12_whitenlearn.npz")
    X = g["X"]
    up = lambda a, b: a * np.sign(np.sum(a * b, axis=1, keepdims=True))
    m, P = whitenlearn(X, g["qidxs"], g["pidxs"], device=DEV)
    np.testing.assert_allclose(m, g["m_lw"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(up(P, g["P_lw"]), g["P_lw"], rtol=1e-7, atol=1e-9)
    m2, P2 = pcawhitenlearn(X, device=DEV)
    np.testing.assert_allclose(m2, g["m_pca"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(up(np.real(P2), g["P_pca"]), g["P_pca"], rtol=1e-7, atol=1e-9)


# ---------------------------------------------------------------- multi-GPU exchange through the C ABI

def test_mdx_comm_single_rank_on_rccl():
    """mdx_comm_* on RCCL with one rank (all a 1-GPU box can hold: RCCL refuses two ranks on one device): the
    communicator comes up, both exchanges deliver the blocks mdx_rank_full_segments expects."""
    from mdir_amd import ops
    rng = np.random.default_rng(3)
    sc = rng.standard_normal((7, 1000)).astype(np.float32)
    comm = ops.Comm(ops.Comm.unique_id(), 1, 0, DEV)
    blocks = comm.allgather_scores(dev(sc), [1000])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(blocks[0].cpu().numpy(), sc)
    mine, (qlo, qhi) = comm.exchange_scores(dev(sc), [1000])
    torch.cuda.synchronize()
    assert (qlo, qhi) == (0, 7)
    np.testing.assert_array_equal(mine[0].cpu().numpy(), sc)
    np.testing.assert_array_equal(ops.rank_full_segments(mine).cpu().numpy(), OC.rank_full(sc))
    comm.close()
    assert ops.query_bounds(70, 8, 0) == (0, 9) and ops.query_bounds(70, 8, 7) == (62, 70)
    assert [ops.query_bounds(70, 8, r) for r in range(8)] == \
        [__import__("mdir_amd.sharded", fromlist=["x"]).query_bounds(70, 8, r) for r in range(8)]


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, text=True,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-4000:])
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two rank processes itself (here both on this
    one GPU over gloo: MDIR_AMD_DRYRUN_ONE_GPU, a functional dry run) and prints rank 0's one line; the sharded
    evaluation gives the single-process mAP."""
    common = ["--rows", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-preflight", "--extract-images", "0"]
    one = _bench(common, {})
    two = _bench(["--gpus", "2"] + common, {"MDIR_AMD_DRYRUN_ONE_GPU": "1"})
    assert two["n_gpus"] == 2 and two["nranks_seen"] == 2
    assert set(two["phases_ms_per_rank"]) >= {"scores", "exchange_exposed", "sort"} and len(two["phases_ms_per_rank"]["scores"]) == 2
    assert two["map_medium"] == one["map_medium"]
    assert "DRY RUN" in two["data"]
    # a node with fewer GPUs than asked for is refused before anything is launched
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], text=True, capture_output=True)
    assert proc.returncode == 2 and "GPU(s)" in proc.stderr


# ---------------------------------------------------------------- the two wave-rank forms, graph capture

_RANK_FORMS_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from mdir_amd import ops
from oracle import chain as OC
dev = "cuda:0"
rng = np.random.default_rng(11)
for n, nq in ((4993, 70), (70000, 5), (300001, 3), (16, 2), (8193, 4)):
    sc = (rng.standard_normal((nq, n)) * 0.03).astype(np.float32)
    m = sc[:, 1::5].shape[1]
    sc[:, 0:5 * m:5] = sc[:, 1::5]                               # ties
    want = OC.rank_full(sc)
    got = ops.rank_full(torch.from_numpy(sc).to(dev)).cpu().numpy()
    assert np.array_equal(got, want), (n, nq)
    k = min(100, n)
    ids, vals = ops.topk(torch.from_numpy(sc).to(dev), k)
    assert np.array_equal(ids.cpu().numpy(), want[:, :k]), (n, nq, "topk")
for kind in ("all_equal", "two_values", "all_nan", "descending"):
    n, nq = 50000, 3
    sc = {"all_equal": np.full((nq, n), 0.25, np.float32), "two_values": rng.choice(np.array([-0.5, 0.5], np.float32), size=(nq, n)),
          "all_nan": np.full((nq, n), np.nan, np.float32), "descending": np.tile(np.linspace(1, -1, n, dtype=np.float32), (nq, 1))}[kind]
    assert np.array_equal(ops.rank_full(torch.from_numpy(sc).to(dev)).cpu().numpy(), OC.rank_full(sc)), kind
print("RANK-FORMS-OK")
"""


@pytest.mark.parametrize("form", ["ballot", "atomic"])
def test_rank_forms(form):
    """Both ways a wave ranks its 64 elements (eight ballots / ds_add_rtn) give the oracle's ranking: the switch is
    read when the library first ranks, so each form gets its own process."""
    env = dict(os.environ, MDX_SORT_RANK=form)
    proc = subprocess.run([sys.executable, "-c", _RANK_FORMS_SCRIPT % {"root": ROOT}], env=env, text=True, capture_output=True, timeout=900)
    assert proc.returncode == 0 and "RANK-FORMS-OK" in proc.stdout, proc.stderr[-3000:]


_CAPTURE_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from mdir_amd import ops
from oracle import chain as OC
dev = "cuda:0"
rng = np.random.default_rng(5)
n, nq = 150000, 4
sc_host = (rng.standard_normal((nq, n)) * 0.03).astype(np.float32)
sc = torch.empty((nq, n), dtype=torch.float32, device=dev)
rk = torch.empty((nq, n), dtype=torch.int64, device=dev)
ws = torch.empty(ops.rank_workspace_bytes(n, nq), dtype=torch.uint8, device=dev)
small = torch.empty((nq, 3000), dtype=torch.float32, device=dev)
rk_small = torch.empty((nq, 3000), dtype=torch.int64, device=dev)
ws_small = torch.empty(ops.rank_workspace_bytes(3000, nq), dtype=torch.uint8, device=dev)
%(warm)s
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        ops.rank_full(sc, out=rk, workspace=ws)                  # the FIRST ranking call of the process when warm is empty
        ops.rank_full(small, out=rk_small, workspace=ws_small)
for rep in range(3):
    sc_host = np.roll(sc_host, 7 + rep, axis=1) * np.float32(1 + rep)
    sc.copy_(torch.from_numpy(sc_host)); small.copy_(torch.from_numpy(np.ascontiguousarray(sc_host[:, :3000])))
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(rk.cpu().numpy(), OC.rank_full(sc_host)), rep
    assert np.array_equal(rk_small.cpu().numpy(), OC.rank_full(np.ascontiguousarray(sc_host[:, :3000]))), rep
print("CAPTURE-OK")
"""


@pytest.mark.parametrize("warm", ["", "ops.rank_full(sc, out=rk, workspace=ws); torch.cuda.synchronize()"])
def test_ranking_under_graph_capture(warm):
    """mdx_rank_full captured into a hipGraph -- as the first ranking call of a process (no index was built, so the
    LDS-order probe has not run: the ballot kernels are captured) and after an eager call -- replays bit-exact."""
    proc = subprocess.run([sys.executable, "-c", _CAPTURE_SCRIPT % {"root": ROOT, "warm": warm}], text=True, capture_output=True, timeout=900)
    assert proc.returncode == 0 and "CAPTURE-OK" in proc.stdout, proc.stderr[-3000:]


_COMM_PATH_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r)
from mdir_amd import ops
from mdir_amd.sharded import ShardedIndex
from oracle import chain as OC
dev = "cuda:0"
rng = np.random.default_rng(2)
n, d, nq = 5000, 64, 9
rows = rng.standard_normal((n, d)).astype(np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
q = np.ascontiguousarray(rows[:nq].T + 0.01)
sh = ShardedIndex(torch.from_numpy(rows).to(dev), "ND", n)
assert sh._comm is not None and sh.chunks == 2
for rep in range(3):
    rk, sc, (qlo, qhi) = sh.rank_queries(torch.from_numpy(q).to(dev), "DN")
    torch.cuda.synchronize()
    want = OC.scores_chain(np.ascontiguousarray(rows.T), q)
    assert (qlo, qhi) == (0, nq) and len(sc.blocks) == 2
    assert np.array_equal(sc.dense().cpu().numpy(), want) and np.array_equal(rk.cpu().numpy(), OC.rank_full(want))
assert sh.phase_ms() is not None
print("COMM-PATH-OK")
"""


def test_sharded_index_through_the_mdx_communicator():
    """MDIR_AMD_COMM=mdx: ShardedIndex moves its blocks with mdx_exchange_scores on a side stream (event-ordered with the
    compute stream) -- exercised here with the one rank a 1-GPU box can hold and two row chunks."""
    env = dict(os.environ, MDIR_AMD_COMM="mdx", MDIR_AMD_EXCHANGE_CHUNKS="2")
    proc = subprocess.run([sys.executable, "-c", _COMM_PATH_SCRIPT % {"root": ROOT}], env=env, text=True, capture_output=True, timeout=600)
    assert proc.returncode == 0 and "COMM-PATH-OK" in proc.stdout, proc.stderr[-3000:]


# ---------------------------------------------------------------- 1x1 convolution with the fused epilogue

@pytest.mark.parametrize("n,cin,cout,h,w,res,relu,bn", [(1, 64, 64, 7, 9, False, True, True), (2, 256, 64, 33, 31, False, True, True),
                                                       (4, 64, 256, 16, 16, True, True, True), (1, 1024, 256, 48, 64, False, True, True),
                                                       (3, 256, 1024, 23, 17, True, False, True), (2, 16, 128, 5, 3, False, False, False),
                                                       (1, 128, 512, 128, 96, True, True, True)])
def test_conv1x1_bn_act_vs_float64(n, cin, cout, h, w, res, relu, bn):
    """mdx_conv1x1_bn_act = relu(bn(conv1x1(x)) + identity) of the torchvision Bottleneck cirtorch keeps
    (imageretrievalnet.py:172-173), against the same arithmetic in float64: fp32 rounding of a Cin-long fma chain."""
    import torch.nn.functional as F
    from mdir_amd import ops
    torch.manual_seed(n + cin + cout)
    x = torch.randn(n, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, 1, 1, device=DEV) / cin ** 0.5
    mean, var = (torch.randn(cout, device=DEV) * 0.1, torch.rand(cout, device=DEV) + 0.5) if bn else (None, None)
    gamma, beta = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.1
    idt = torch.randn(n, cout, h, w, device=DEV) if res else None
    got = ops.conv1x1_bn_act(x, ops.conv1x1_transpose_weights(wt), mean, var, gamma, beta, 1e-5, idt, relu)
    y = F.conv2d(x.double(), wt.double())
    scale = gamma.double() / torch.sqrt(var.double() + 1e-5) if bn else gamma.double()
    y = (y - (mean.double() if bn else torch.zeros(cout, device=DEV, dtype=torch.float64)).view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1) \
        + beta.double().view(1, -1, 1, 1)
    if res:
        y = y + idt.double()
    if relu:
        y = y.clamp_min(0)
    assert float((got.double() - y).abs().max() / y.abs().max()) < 3e-6
    with pytest.raises(ValueError):
        ops.conv1x1_bn_act(x, ops.conv1x1_transpose_weights(wt)[:, :cout - 1].contiguous(), None, None)      # Cout % 64


def test_trunk_with_own_1x1_convolutions_equals_library_trunk(monkeypatch):
    """ResNet50 with its 1x1 convolutions on mdx_conv1x1_bn_act -- all of them (MDIR_AMD_CONV1X1=1) or the default subset
    (auto: the expand convolutions and those with <= 64 output channels) -- equals the all-MIOpen trunk (=0) to fp32 rounding."""
    from mdir_amd.backbones import TrunkSequential, build_features
    torch.manual_seed(5)
    feats = TrunkSequential(*build_features("resnet50")).eval()
    for m in feats.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.2); m.bias.data.normal_(0, 0.1)
    feats = feats.to(DEV)
    x = torch.randn(2, 3, 203, 157, device=DEV)
    with torch.no_grad():
        monkeypatch.setenv("MDIR_AMD_CONV1X1", "0")
        plain = feats(x)
        monkeypatch.setenv("MDIR_AMD_CONV1X1", "1")
        own = feats(x)
        monkeypatch.delenv("MDIR_AMD_CONV1X1")
        auto = feats(x)
    for got in (own, auto):
        assert float((got - plain).abs().max()) <= 2e-5 * float(plain.abs().max())
        assert not torch.equal(got, plain)               # a different summation order: the own kernels did run
    assert not torch.equal(own, auto)                    # ... and `auto` is a proper subset


# ---------------------------------------------------------------- CLAHE pre-processing (f4; parity unpinned against OpenCV)

@pytest.mark.parametrize("b,h,w,clip,grid", [(1, 64, 96, 4, 8), (2, 37, 53, 4, 8), (1, 240, 320, 2, (4, 6)), (1, 60, 96, 4, 8),
                                             (1, 9, 7, 4, 8), (3, 128, 128, 0, 8), (1, 768, 1024, 4, 8)])
def test_clahe_kernels_vs_restatement(b, h, w, clip, grid):
    """mdx_clahe_u8_to_chw against oracle.apply_clahe_rgb (the same restatement of OpenCV's algorithm in numpy): the uint8
    lightness may differ by one level where powf / cbrtf differ in the last bit at a truncation boundary (< 0.1 % of the
    pixels); GIVEN the device's lightness plane the per-tile LUTs and the equalised lightness are bit-exact and the output is
    within 1e-4 (normalised units: powf / cbrtf differ from numpy in the last bits, and 1 / std amplifies 4.4x)."""
    from oracle import oracle as O
    from mdir_amd import ops
    rng = np.random.default_rng(h * w + b)
    base = rng.integers(0, 256, (b, h // 8 + 1, w // 8 + 1, 3))
    img = np.clip(np.kron(base, np.ones((1, 8, 8, 1)))[:, :h, :w] + rng.normal(0, 12, (b, h, w, 3)), 0, 255).astype(np.uint8)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    out, l8, luts, l8_eq = ops.clahe_u8_to_chw(dev(img), clip, grid, mean, std, return_intermediates=True)
    out, l8, luts, l8_eq = out.cpu().numpy(), l8.cpu().numpy(), luts.cpu().numpy(), l8_eq.cpu().numpy()
    g = grid if isinstance(grid, tuple) else (grid, grid)
    for i in range(b):
        want_rgb, want_l8 = O.apply_clahe_rgb(img[i], clip, g)
        diff = l8[i].astype(int) - want_l8.astype(int)
        assert np.abs(diff).max() <= 1 and (diff != 0).mean() < 1e-3
        want_luts, tile = O.clahe_luts(l8[i], clip, g)
        np.testing.assert_array_equal(luts[i], want_luts)
        np.testing.assert_array_equal(l8_eq[i], O.clahe_apply(l8[i], want_luts, tile))     # the blend, ties included: bit-exact
        # the rest of the chain on the device's own lightness plane
        lab = O.rgb_to_lab(img[i].astype(np.float32) / np.float32(255))
        spc = ((lab + np.array([0, 128, 128], np.float32)) / np.array([100, 255, 255], np.float32)).astype(np.float32)
        spc[..., 0] = O.clahe_apply(l8[i], want_luts, tile).astype(np.float32) / np.float32(255)
        rgb = O.lab_to_rgb(((spc * np.array([100, 255, 255], np.float32)).astype(np.float32) - np.array([0, 128, 128], np.float32)).astype(np.float32))
        want = ((rgb - np.float32(mean)) / np.float32(std)).transpose(2, 0, 1)
        np.testing.assert_allclose(out[i], want, rtol=0, atol=1e-4)
        if (diff == 0).all():
            np.testing.assert_allclose(out[i], ((want_rgb - np.float32(mean)) / np.float32(std)).transpose(2, 0, 1), rtol=0, atol=1e-4)


def test_extraction_through_the_clahe_chain(tmp_path, monkeypatch):
    """extract_vectors with the transform chain the CLAHE networks' checkpoints carry: file -> JPEG decode -> thumbnail -> CLAHE
    + normalise (all on the device) -> network; equal to the same network fed the restated chain's tensors one by one."""
    from PIL import Image
    from oracle import oracle as O
    from mdir_amd.datasets import initialize_transforms
    from mdir_amd.networks import extract_vectors, init_network
    monkeypatch.setenv("MDIR_AMD_WORKERS", "2")
    rng = np.random.default_rng(8)
    names = []
    for i, (w, h) in enumerate(((320, 240), (240, 320), (320, 240))):
        pic = np.clip(np.kron(rng.integers(0, 255, (h // 16, w // 16, 3)), np.ones((16, 16, 1))) + rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8)
        names.append(str(tmp_path / ("im%d.jpg" % i)))
        Image.fromarray(pic).save(names[-1], quality=90)
    torch.manual_seed(3)
    net = init_network({"architecture": "alexnet", "pooling": "gem", "whitening": False, "pretrained": False}).eval().to(DEV)
    tr = initialize_transforms("pil2np | apply_clahe | totensor | normalize", [net.meta["mean"], net.meta["std"]])
    with torch.no_grad():
        got = extract_vectors(net, names, 256, tr, device=torch.device(DEV)).numpy()       # [D, N]
        for i, path in enumerate(names):
            img = Image.open(path).convert("RGB")
            img.thumbnail((256, 256), Image.LANCZOS)
            rgb, _ = O.apply_clahe_rgb(np.array(img), 4, 8)
            x = torch.from_numpy(((rgb - np.float32(net.meta["mean"])) / np.float32(net.meta["std"])).transpose(2, 0, 1)[None].copy()).to(DEV)
            np.testing.assert_allclose(got[:, i], net(x).cpu().numpy().reshape(-1), atol=2e-4)
