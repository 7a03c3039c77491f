"""Parity of the HIP kernels (through the C ABI) against the CPU oracle.

Bar: bit-exact for ids / positions; scores bit-exact against the fmaf-chain oracle
(oracle/chain.c) and within 1e-5 of the reference's np.dot; pooled / normalised
floats within rtol 1e-5 (tolerance of BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from conftest import sparse_map
from oracle import chain as OC
from oracle import oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from mdir_amd import ops as _ops
    return _ops


def _unit_rows(rng, n, d):
    v = rng.standard_normal((n, d)).astype(np.float32)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


# shards of >= 32 768 rows take 128-row workgroups (R = 2), smaller ones 64-row workgroups; with R = 2 a last
# query tile of <= 8 queries goes through the v_mfma_f32_4x4x1 leftover path (mdx_scores_kernel.h)
@pytest.mark.parametrize("n,d,nq", [(4993, 2048, 70), (6322, 2048, 70), (1000, 512, 1), (333, 100, 17), (16, 64, 16),
                                    (5000, 256, 130), (70, 2048, 70),
                                    # leftover path: 1, 2, 6 and 7 full tiles + a last tile of 8, 1, 4 and 8 queries
                                    (40000, 128, 24), (32768, 32, 33), (33000, 64, 100), (50000, 48, 120),
                                    (70000, 64, 1), (66001, 100, 17), (70001, 256, 130), (65600, 2048, 70),
                                    (65537, 32, 128), (131072, 96, 33),
                                    # > 256 queries: the full groups of 128 go out as ONE launch (grid.y = group)
                                    (1125, 512, 1125), (3000, 128, 300), (40000, 64, 389), (2000, 256, 256)])
def test_scores_bit_exact_vs_chain(ops, n, d, nq):
    rng = np.random.default_rng(n + d + nq)
    db, qv = _unit_rows(rng, n, d), _unit_rows(rng, nq, d)
    vecs, qvecs = np.ascontiguousarray(db.T), np.ascontiguousarray(qv.T)   # reference layout [D,N],[D,Q]
    want = OC.scores_chain(vecs, qvecs)
    ix = ops.DescriptorIndex(dev(vecs), "DN")
    got = ix.scores(dev(qvecs), "DN").cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # row-major inputs give the same bits
    ix2 = ops.DescriptorIndex(dev(db), "ND")
    got2 = ix2.scores(dev(qv), "ND").cpu().numpy()
    np.testing.assert_array_equal(got2, want)
    # and the reference's BLAS result is within the north-star tolerance
    np.testing.assert_allclose(got.T, O.scores(vecs, qvecs), rtol=0, atol=1e-5)


def test_scores_golden_small(ops, golden):
    g = golden("g7_ranking.npz")
    for name in "abc":
        ix = ops.DescriptorIndex(dev(g[f"{name}_vecs"]), "DN")
        sc = ix.scores(dev(g[f"{name}_qvecs"]), "DN")
        np.testing.assert_allclose(sc.cpu().numpy().T, g[f"{name}_scores"], rtol=0, atol=1e-6)
        rk = ops.rank_full(sc).cpu().numpy()
        np.testing.assert_array_equal(rk.T, g[f"{name}_ranks"])


def test_scores_center(ops):
    rng = np.random.default_rng(3)
    P, X = rng.standard_normal((96, 80)).astype(np.float32), rng.standard_normal((9, 80)).astype(np.float32)
    m = rng.standard_normal(80).astype(np.float32)
    ix = ops.DescriptorIndex(dev(P), "ND")
    got = ix.scores(dev(X), "ND", center=dev(m)).cpu().numpy()
    np.testing.assert_array_equal(got, OC.gemm_nt_chain(X - m, P))


@pytest.mark.parametrize("n,nq", [(4993, 70), (6322, 70), (1, 1), (63, 3), (4096, 2), (4097, 2), (70000, 5)])
def test_rank_full_bit_exact(ops, n, nq):
    rng = np.random.default_rng(n * 7 + nq)
    sc = rng.standard_normal((nq, n)).astype(np.float32)
    if n > 100:
        sc[:, 10:40] = sc[:, 5:6]      # runs of exact ties
        sc[0, 50] = np.nan
        sc[0, 51] = -0.0
        sc[0, 52] = 0.0
    want = OC.rank_full(sc)
    got = ops.rank_full(dev(sc)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    got = ops.rank_full(dev(sc), id_offset=1000).cpu().numpy()
    np.testing.assert_array_equal(got, want + 1000)


@pytest.mark.parametrize("widths", [[2500, 2493], [4096, 4096, 1], [1, 5000, 0, 3, 9000], [700] * 32, [4993],
                                    [125625, 125624, 125624]])
def test_rank_full_segments_equals_rank_full(ops, widths):
    """Scores delivered as column blocks (the peer blocks of the multi-GPU exchange) enter the sort in place: same
    ranking as the concatenated matrix, bit for bit -- blocks that end inside a sort tile, empty blocks, one block,
    the maximum of 32, ties across block borders, NaN."""
    rng = np.random.default_rng(sum(widths))
    nq = 3
    blocks = [(np.round(rng.standard_normal((nq, w)) * 50) / 50).astype(np.float32) for w in widths]
    blocks[0][1, 0] = np.nan
    whole = np.concatenate(blocks, axis=1)
    got = ops.rank_full_segments([dev(b) for b in blocks], id_offset=11).cpu().numpy()
    np.testing.assert_array_equal(got, OC.rank_full(whole) + 11)
    np.testing.assert_array_equal(got, ops.rank_full(dev(whole), id_offset=11).cpu().numpy())
    with pytest.raises(ValueError):
        ops.rank_full_segments([dev(blocks[0])] * 33)


def test_rank_tie_fixture(ops, golden):
    g = golden("g7_ranking.npz")
    ix = ops.DescriptorIndex(dev(g["tie_vecs"]), "DN")
    sc = ix.scores(dev(g["tie_qvecs"]), "DN")
    np.testing.assert_array_equal(sc.cpu().numpy().T, g["tie_scores"])
    rk = ops.rank_full(sc).cpu().numpy().T
    np.testing.assert_array_equal(rk, O.ranks(g["tie_scores"]))


def test_topk_and_rank_of(ops):
    rng = np.random.default_rng(11)
    nq, n = 6, 20000
    sc = rng.standard_normal((nq, n)).astype(np.float32)
    sc[:, 100:164] = sc[:, 99:100]
    full = OC.rank_full(sc)
    ids, vals = ops.topk(dev(sc), 100)
    np.testing.assert_array_equal(ids.cpu().numpy(), full[:, :100])
    np.testing.assert_array_equal(vals.cpu().numpy(), np.take_along_axis(sc, full[:, :100], axis=1))
    lists = [rng.choice(n, size=s, replace=False) for s in (5, 0, 300, 1, 64, 17)]
    lists[2][:10] = np.arange(100, 110)   # inside the tie run
    pos, idsc, off = ops.rank_of(dev(sc), lists)
    pos, idsc = pos.cpu().numpy(), idsc.cpu().numpy()
    for q in range(nq):
        np.testing.assert_array_equal(pos[off[q]:off[q + 1]], OC.rank_of(sc[q], lists[q]))
        np.testing.assert_array_equal(idsc[off[q]:off[q + 1]], sc[q][lists[q]])


def test_rank_of_many_labelled_ids(ops):
    """mdx_rank_of with more labelled ids than one sweep of the counting kernel sorts (256), duplicates in
    the list, an empty list, ties and NaN scores: positions == the inverse of the oracle ranking."""
    rng = np.random.default_rng(5)
    n, nq = 50000, 4
    s = (np.round(rng.standard_normal((nq, n)) * 40) / 40).astype(np.float32)        # heavy ties
    s[2, ::7] = np.nan
    lists = [rng.choice(n, 700, replace=False), np.array([5, 5, 9, 123, 9]), np.array([], dtype=np.int64),
             rng.choice(n, 257, replace=False)]
    pos, sc, off = ops.rank_of(dev(s), lists)
    pos = pos.cpu().numpy()
    want = OC.rank_full(s)
    inv = np.empty_like(want)
    np.put_along_axis(inv, want, np.broadcast_to(np.arange(n), want.shape), axis=1)
    for q in range(nq):
        np.testing.assert_array_equal(pos[off[q]:off[q + 1]], inv[q][lists[q]])


def test_pool_l2n_golden(ops, golden):
    g = golden("g1_pool.npz")
    for c, h, w in [(2048, 24, 32), (2048, 17, 23), (512, 48, 64), (256, 7, 5)]:
        x = sparse_map(int(g[f"seed_c{c}_h{h}_w{w}"]), (1, c, h, w))
        xd = dev(x)
        for p in (3.0, 2.2, 1.0):
            got = ops.pool_l2n(xd, "gem", p, l2n_eps=None).cpu().numpy()[0]
            np.testing.assert_allclose(got, g[f"gem_c{c}_h{h}_w{w}_p{p}"], rtol=1e-5, atol=1e-7)
            both = ops.pool_l2n(xd, "gem", p).cpu().numpy()
            np.testing.assert_allclose(both, O.l2n(O.gem(x, p)), rtol=1e-5, atol=1e-7)
        np.testing.assert_array_equal(ops.pool_l2n(xd, "mac", l2n_eps=None).cpu().numpy()[0], g[f"mac_c{c}_h{h}_w{w}"])
        np.testing.assert_allclose(ops.pool_l2n(xd, "spoc", l2n_eps=None).cpu().numpy()[0], g[f"spoc_c{c}_h{h}_w{w}"],
                                   rtol=1e-5)


def test_pool_batch_and_zero_map(ops):
    x = sparse_map(5, (3, 64, 6, 9))
    x[1] = 0.0
    got = ops.pool_l2n(dev(x), "gem", 2.92).cpu().numpy()
    np.testing.assert_allclose(got, O.l2n(O.gem(x, 2.92)), rtol=1e-5, atol=1e-7)
    assert not np.isnan(got).any()


@pytest.mark.parametrize("B,C,sizes", [(1, 2048, [(24, 32), (17, 23), (12, 16)]), (4, 512, [(48, 64), (33, 45), (24, 32)]),
                                       (3, 67, [(7, 5), (1, 1)]), (2, 256, [(6, 9)] * 8)])
@pytest.mark.parametrize("kind,p", [("gem", 3.0), ("gem", 2.92), ("mac", 1.0), ("spoc", 1.0)])
def test_pool_multi_l2n_aggregate_two_launch_tail(ops, B, C, sizes, kind, p):
    """The two-launch descriptor tail (mdx_pool_multi + mdx_l2n_aggregate) of a pyramid: bit-identical to the launches
    it replaces (mdx_pool_l2n per scale, mdx_ms_aggregate_batch) and equal to the oracle's gem -> l2n -> aggregate."""
    maps = [sparse_map(20 + i, (B, C, h, w)) for i, (h, w) in enumerate(sizes)]
    maps[0][B - 1] = 0.0                                            # an all-zero map: eps keeps its row finite
    md = [dev(m) for m in maps]
    pooled = ops.pool_multi(md, kind, p)
    assert pooled.shape == (len(sizes), B, C)
    for s_, m in enumerate(md):
        np.testing.assert_array_equal(pooled[s_].cpu().numpy(), ops.pool_l2n(m, kind, p, l2n_eps=None).cpu().numpy())
    msp = p if kind == "gem" else 1.0
    got = ops.l2n_aggregate(pooled, 1e-6, msp).cpu().numpy()
    want_launches = ops.ms_aggregate_batch([ops.pool_l2n(m, kind, p) for m in md], msp).cpu().numpy()
    np.testing.assert_array_equal(got, want_launches)
    pool = {"gem": lambda x: O.gem(x, p), "mac": O.mac, "spoc": O.spoc}[kind]
    per = np.stack([O.l2n(pool(m)) for m in maps])                  # [S,B,C]
    live = [b for b in range(B) if np.abs(per[:, b]).sum() > 0]     # 0/0 in the renormalisation of an all-zero image (as the reference)
    want = np.stack([O.ms_aggregate(per[:, b], msp) for b in live])
    np.testing.assert_allclose(got[live], want, rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        ops.pool_multi([md[0], md[0][:, :C - 1].contiguous()], kind, p)


@pytest.mark.parametrize("shape", [(1, 64, 48, 64), (2, 256, 17, 23), (1, 2048, 23, 17), (3, 5, 1, 1), (1, 7, 3, 5),
                                   (1, 64, 384, 512), (1, 2048, 24, 32)])
def test_bn_act_vs_oracle_and_torch(ops, shape):
    """Fused trunk epilogue: every combination of residual / relu / affine, aligned and odd planes,
    against the float64 oracle and against torch's own batch_norm + add + relu on the same device."""
    rng = np.random.default_rng(11)
    n, c, h, w = shape
    x = rng.standard_normal(shape).astype(np.float32) * 2
    res = rng.standard_normal(shape).astype(np.float32)
    mean, var = rng.standard_normal(c).astype(np.float32), rng.uniform(0.2, 3.0, c).astype(np.float32)
    wt, bs = rng.uniform(0.5, 1.5, c).astype(np.float32), rng.standard_normal(c).astype(np.float32)
    for use_res in (False, True):
        for relu in (False, True):
            for affine in (True, False):
                got = ops.bn_act_(dev(x.copy()), dev(mean), dev(var), dev(wt) if affine else None, dev(bs) if affine else None,
                                  1e-5, dev(res) if use_res else None, relu)
                want = O.bn_act(x, mean, var, wt if affine else None, bs if affine else None, 1e-5, res if use_res else None, relu)
                np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=2e-6)
                t = torch.nn.functional.batch_norm(dev(x), dev(mean), dev(var), dev(wt) if affine else None,
                                                   dev(bs) if affine else None, False, 0.0, 1e-5)
                if use_res:
                    t = t + dev(res)
                if relu:
                    t = torch.relu(t)
                np.testing.assert_allclose(got.cpu().numpy(), t.cpu().numpy(), rtol=2e-6, atol=2e-6)
    with pytest.raises(ValueError):
        ops.bn_act_(dev(x), dev(mean[:-1]) if c > 1 else dev(np.zeros(2, np.float32)), dev(var))


@pytest.mark.parametrize("w,h,imsize,batch", [(1600, 1200, 1024, 1), (1200, 1600, 1024, 2), (221, 150, 64, 3), (97, 203, 64, 1),
                                              (1025, 700, 1024, 1), (2047, 33, 1024, 1), (640, 480, 362, 4), (3000, 2000, 1024, 1)])
def test_device_thumbnail_is_pillow(ops, w, h, imsize, batch):
    """mdx_resample_u8 (width pass, height pass) with the host's fixed-point LANCZOS taps = Pillow's
    ``Image.thumbnail((imsize, imsize), LANCZOS)`` -- what the reference's imresize calls (datahelpers.py:48-50) --
    pixel for pixel; also against the oracle's restatement."""
    from PIL import Image
    from mdir_amd.resample import DeviceThumbnail, on_device
    rng = np.random.default_rng(w + h)
    arr = rng.integers(0, 256, (batch, h, w, 3), dtype=np.uint8)
    arr[0, : h // 2] = (arr[0, : h // 2] // 64) * 85                 # flat areas and hard edges (ringing is clipped at 0 / 255)
    assert on_device(w, h, imsize) is not None
    got = DeviceThumbnail(imsize)(torch.from_numpy(arr).to(DEV)).cpu().numpy()
    for b in range(batch):
        im = Image.fromarray(arr[b])
        im.thumbnail((imsize, imsize), Image.LANCZOS)
        np.testing.assert_array_equal(got[b], np.asarray(im))
    if w * h < 10 ** 6:
        np.testing.assert_array_equal(got[0], O.thumbnail_u8(arr[0], imsize))


@pytest.mark.parametrize("shape", [(1, 3, 768, 1024), (2, 3, 543, 724), (1, 3, 97, 61), (3, 1, 8, 5)])
def test_bilinear_pyramid_is_f_interpolate(ops, shape):
    """mdx_bilinear_pyramid = F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=False) (wrapper.py:104-107,
    imageretrievalnet.py:315) for the eval pyramid [1, 1/sqrt 2, 1/2] and odd scales: same output sizes, values within one
    rounding of torch's kernel on the same device and of its CPU kernel (golden G6 / G10 go through it in test_gpu_api)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(shape[2])
    x = rng.standard_normal(shape).astype(np.float32)
    xd = dev(x)
    scales = [1, 1. / np.sqrt(2), 1. / 2, 0.3, 0.9]
    if min(shape[2:]) * 0.3 < 1:
        scales = scales[:3]
    got = ops.bilinear_pyramid(xd, scales)
    assert got[0] is xd
    for s_, g in zip(scales[1:], got[1:]):
        want_dev = F.interpolate(xd, scale_factor=s_, mode="bilinear", align_corners=False)
        want_cpu = F.interpolate(torch.from_numpy(x), scale_factor=s_, mode="bilinear", align_corners=False)
        assert g.shape == want_dev.shape == want_cpu.shape
        np.testing.assert_allclose(g.cpu().numpy(), want_dev.cpu().numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(g.cpu().numpy(), want_cpu.numpy(), rtol=0, atol=2e-6)


def test_jpeg_pixels_is_pillow(ops):
    """The device half of the JPEG decoder (mdx_jpeg_pixels: dequantisation, islow IDCT, fancy upsampling, YCbCr -> RGB)
    on the coefficients the host half delivers = Pillow's ``Image.open(f).convert('RGB')`` (datahelpers.py:24-31), pixel
    for pixel: 4:4:4 / 4:2:2 / 4:2:0 / grey, odd sizes, qualities 1..100, saturated pictures, restart markers, progressive
    files, a crop box."""
    import io
    from PIL import Image
    from mdir_amd import jpeg
    from test_oracle_golden import _jpeg_cases
    cases, picture = _jpeg_cases()
    buf = io.BytesIO()
    picture(1600, 1200, "photo").save(buf, format="JPEG", quality=90)
    cases.append(("camera-sized", buf.getvalue()))
    for name, data in cases:
        item = jpeg.entropy_decode(data)
        assert item is not None, name
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        got = jpeg.pixels(item.pin_memory(), torch.device(DEV))
        assert got.shape == (1,) + want.shape
        np.testing.assert_array_equal(got[0].cpu().numpy(), want, err_msg=name)
    item = jpeg.entropy_decode(cases[2][1], box=(3, 5, 100, 70))
    want = np.asarray(Image.open(io.BytesIO(cases[2][1])).convert("RGB").crop((3, 5, 100, 70)))
    np.testing.assert_array_equal(jpeg.pixels(item, torch.device(DEV))[0].cpu().numpy(), want)
    buf = io.BytesIO()
    picture(64, 64, "noise").convert("CMYK").save(buf, format="JPEG")
    assert jpeg.entropy_decode(buf.getvalue()) is None and jpeg.entropy_decode(cases[12][1][:5000]) is None


def test_u8_to_chw_matches_host_chain(ops):
    """mdx_u8_to_chw == `pil2np | totensor | normalize` bit for bit (RGB and single channel, odd sizes, batch)."""
    rng = np.random.default_rng(2)
    for shape, mean, std in (((2, 37, 53, 3), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]), ((1, 5, 3, 1), [0.5], [0.25]),
                             ((1, 768, 1024, 3), [0.1, 0.2, 0.3], [1.0, 0.5, 2.0])):
        u8 = rng.integers(0, 256, shape, dtype=np.uint8)
        got = ops.u8_to_chw(dev(u8), mean, std).cpu().numpy()
        want = ((u8.astype(np.float32) / np.float32(255.0)) - np.array(mean, np.float32)) / np.array(std, np.float32)
        np.testing.assert_array_equal(got, want.transpose(0, 3, 1, 2))
    with pytest.raises(ValueError):
        ops.u8_to_chw(dev(u8), [0.0], [1.0])
    with pytest.raises(ValueError):
        ops.u8_to_chw(dev(np.zeros((1, 2, 2, 3), np.uint8)), [0, 0, 0], [1, 0, 1])       # zero std


def test_fused_trunk_equals_module_calls(ops, monkeypatch):
    """ResNet blocks with the fused epilogue == the same blocks through torch's bn / add / relu kernels."""
    from mdir_amd.backbones import build_features
    torch.manual_seed(5)
    from mdir_amd.backbones import TrunkSequential
    feats = TrunkSequential(*build_features("resnet50")).eval()
    for m in feats.modules():
        if isinstance(m, torch.nn.BatchNorm2d):       # non-trivial running statistics
            m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.2); m.bias.data.normal_(0, 0.1)
    feats = feats.to(DEV)
    x = torch.randn(1, 3, 203, 157, device=DEV)
    with torch.no_grad():
        fused = feats(x)
        monkeypatch.setenv("MDIR_AMD_FUSED_TRUNK", "0")
        plain = feats(x)
    assert fused.shape == plain.shape
    scale = float(plain.abs().max())
    assert float((fused - plain).abs().max()) <= 2e-5 * scale, (float((fused - plain).abs().max()), scale)
    # autograd keeps the module path (no in-place HIP kernel under grad)
    monkeypatch.delenv("MDIR_AMD_FUSED_TRUNK")
    y = feats(x.requires_grad_(True))
    y.mean().backward()
    assert x.grad is not None
    # VGG / AlexNet layers: conv bias + ReLU folded
    vgg = TrunkSequential(*build_features("vgg16")).eval().to(DEV)
    x = torch.randn(2, 3, 97, 131, device=DEV)
    with torch.no_grad():
        fused = vgg(x)
        monkeypatch.setenv("MDIR_AMD_FUSED_TRUNK", "0")
        plain = vgg(x)
    assert float((fused - plain).abs().max()) <= 2e-5 * float(plain.abs().max())
    # bias-only form of the kernel is exactly x + b
    b = torch.randn(7, device=DEV)
    t = torch.randn(1, 7, 5, 3, device=DEV)
    np.testing.assert_array_equal(ops.bn_act_(t.clone(), None, None, None, b, 0.0, None, False).cpu().numpy(),
                                  (t + b.view(1, -1, 1, 1)).cpu().numpy())


def test_l2n_rows_golden(ops, golden):
    g = golden("g2_l2n.npz")
    got = ops.l2n_rows_(dev(g["x"].copy())).cpu().numpy()
    np.testing.assert_allclose(got, g["y"], rtol=1e-6, atol=1e-9)
    assert np.all(got[2] == 0)


def test_forward_tail_golden(ops, golden):
    """pool -> L2N -> nn.Linear whitening -> L2N (imageretrievalnet.py:107-115)."""
    g = golden("g3_tail.npz")
    feat = dev(g["feat"])
    wix = ops.DescriptorIndex(dev(g["w"]), "ND")
    for p in (3.0, 2.92):
        o = ops.pool_l2n(feat, "gem", p)
        np.testing.assert_allclose(o.cpu().numpy().T, g[f"out_plain_p{p}"], rtol=1e-5, atol=1e-7)
        y = wix.scores(o, "ND")                       # [B, D] = o @ W^T
        ops.l2n_rows_(y, bias=dev(g["b"]))
        np.testing.assert_allclose(y.cpu().numpy().T, g[f"out_whiten_p{p}"], rtol=1e-5, atol=2e-7)


def test_ms_aggregate_golden(ops, golden):
    g = golden("g4_aggregate.npz")
    vs = [dev(v) for v in g["vecs"]]
    for msp in (1.0, 3.0, 2.92):
        got = ops.ms_aggregate(vs, msp).cpu().numpy()
        np.testing.assert_allclose(got, g[f"agg_msp{msp}"], rtol=1e-5, atol=1e-8)


def test_ms_aggregate_batch_golden(ops, golden):
    """Batched aggregation (one launch, a workgroup per image): every row equals the per-image kernel and golden G4."""
    g = golden("g4_aggregate.npz")
    S, D = g["vecs"].shape
    rng = np.random.default_rng(2)
    other = np.abs(rng.standard_normal((S, D))).astype(np.float32)
    other /= np.linalg.norm(other, axis=1, keepdims=True)
    mats = [dev(np.stack([g["vecs"][s], other[s], g["vecs"][s]])) for s in range(S)]          # [B=3, D] per scale
    for msp in (1.0, 3.0, 2.92):
        got = ops.ms_aggregate_batch(mats, msp).cpu().numpy()
        np.testing.assert_allclose(got[0], g[f"agg_msp{msp}"], rtol=1e-5, atol=1e-8)
        np.testing.assert_array_equal(got[0], got[2])
        np.testing.assert_array_equal(got[1], ops.ms_aggregate([dev(other[s]) for s in range(S)], msp).cpu().numpy())


def test_whiten_golden(ops, golden):
    """CirtorchWhiten.postprocess as index-of-P x centred descriptors (wrapper.py:193-195)."""
    g = golden("g5_whiten.npz")
    P32, m32 = g["P"].astype(np.float32), g["m"].astype(np.float32).reshape(-1)
    X = g["X"]
    for dims in (None, 48):
        d = dims or P32.shape[0]
        pix = ops.DescriptorIndex(dev(P32[:d]), "ND")
        y = pix.scores(dev(X), "DN", center=dev(m32))      # [N, d]
        np.testing.assert_array_equal(y.cpu().numpy(), OC.gemm_nt_chain((X.T - m32), P32[:d]))
        ops.l2n_rows_(y, eps=1e-6)
        np.testing.assert_allclose(y.cpu().numpy().T, g[f"wrapper_dims{dims}"], rtol=1e-5, atol=2e-7)


def test_errors_are_loud(ops):
    with pytest.raises(RuntimeError):
        ops.pool_l2n(torch.zeros(1, 4, 2, 2), "gem")           # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.pool_l2n(torch.zeros(1, 4, 2, 2, device=DEV), "gem", p=-1.0)
    ix = ops.DescriptorIndex(torch.zeros(8, 40, device=DEV), "DN")
    with pytest.raises(ValueError):
        ix.scores(torch.zeros(9, 3, device=DEV), "DN")
    with pytest.raises(IndexError):
        ops.rank_of(torch.zeros(1, 10, device=DEV), [[10]])


def test_scores_race_screen(ops):
    """The loader/consumer ring under repetition and uneven load: random shapes, four launches each
    with other work on a second stream, every score bit-exact (tools/stress_scores.py is the long form)."""
    rng = np.random.default_rng(123)
    bg = torch.cuda.Stream()
    junk = torch.randn(2048, 2048, device=DEV)
    for it in range(10):
        n = int(rng.integers(1, 90000)) if it % 2 else int(rng.integers(40000, 120000))
        d = int(rng.choice([32, 100, 512, 2048]))
        nq = int(rng.integers(1, 140))
        db = (rng.standard_normal((n, d)) / np.sqrt(d)).astype(np.float32)
        q = (rng.standard_normal((nq, d)) / np.sqrt(d)).astype(np.float32)
        want = OC.gemm_nt_chain(q, db)
        ix = ops.DescriptorIndex(dev(db), "ND")
        qd = dev(q)
        for rep in range(4):
            with torch.cuda.stream(bg):
                _ = junk @ junk
            np.testing.assert_array_equal(ix.scores(qd, "ND").cpu().numpy(), want)
        ix.close()


@pytest.mark.parametrize("kind", ["all_equal", "ascending", "descending", "two_values", "all_nan", "few_distinct", "denormals"])
def test_rank_full_degenerate_distributions(ops, kind):
    """Histograms with one huge bin, already-sorted input, ties everywhere: still the exact stable order."""
    rng = np.random.default_rng(1)
    n, nq = 50000, 3
    if kind == "all_equal":
        sc = np.full((nq, n), 0.25, dtype=np.float32)
    elif kind == "ascending":
        sc = np.tile(np.linspace(-1, 1, n, dtype=np.float32), (nq, 1))
    elif kind == "descending":
        sc = np.tile(np.linspace(1, -1, n, dtype=np.float32), (nq, 1))
    elif kind == "two_values":
        sc = rng.choice(np.array([-0.5, 0.5], dtype=np.float32), size=(nq, n))
    elif kind == "all_nan":
        sc = np.full((nq, n), np.nan, dtype=np.float32)
    elif kind == "few_distinct":
        sc = rng.choice(rng.standard_normal(37).astype(np.float32), size=(nq, n))
    else:
        sc = (rng.standard_normal((nq, n)) * 1e-41).astype(np.float32)      # subnormal scores, both signs, zeros
        sc[:, ::7] = 0.0
        sc[:, 3::11] = -0.0
    want = OC.rank_full(sc)
    got = ops.rank_full(dev(sc)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    if kind in ("all_equal", "all_nan"):
        np.testing.assert_array_equal(got[0], np.arange(n))
    ids, _ = ops.topk(dev(sc), 257)
    np.testing.assert_array_equal(ids.cpu().numpy(), want[:, :257])


@pytest.mark.parametrize("n,nq,k,kind", [(200000, 5, 100, "gauss"), (70000, 3, 1, "gauss"), (65536, 4, 1000, "ties"),
                                         (100000, 2, 50, "allequal"), (300000, 3, 257, "concentrated"),
                                         (50000, 2, 10, "nan"), (20000, 6, 100, "gauss"), (1000, 3, 10, "gauss"),
                                         (40000, 2, 5000, "gauss"), (300000, 4, 100, "clustered"),
                                         (1004993, 3, 100, "gauss"), (400000, 2, 1000, "sortedrows"),
                                         (262144, 3, 64, "mostlynan"), (300000, 3, 100, "adversarial")])
def test_topk_radix_select(ops, n, nq, k, kind):
    """mdx_topk (sampled threshold for k <<< n, radix select for k << n, trimmed full sort otherwise)
    = first k of the oracle ranking.  "clustered" puts all the high scores into two tiles (per-tile
    candidate slots overflow into the query's spill list); "adversarial" makes exactly the rows the
    sampled path looks at score LOW, so that its threshold admits nearly every row and the exact
    in-kernel fallback has to produce the answer; "mostlynan" leaves few finite scores."""
    rng = np.random.default_rng(n + k)
    sc = (rng.standard_normal((nq, n)) * 0.022).astype(np.float32)
    if kind == "ties":
        sc = (np.round(sc * 50) / 50).astype(np.float32)
    if kind == "allequal":
        sc[:] = 0.125
    if kind == "concentrated":
        sc = (0.3 + rng.standard_normal((nq, n)) * 1e-4).astype(np.float32)
    if kind == "nan":
        sc[:, ::3] = np.nan
    if kind == "clustered":
        sc[:, 1000:7000] += 1.0
    if kind == "sortedrows":
        sc = -np.sort(-sc, axis=1)                   # score descending with the row id
    if kind == "adversarial":                        # the sample positions of tks_sample_kernel, restated
        stride = n // 4096
        j = np.arange(4096, dtype=np.uint64)
        for q in range(nq):
            jitter = (((j * 2654435761) & 0xFFFFFFFF) ^ ((q * 40503 + 0x9E3779B9) & 0xFFFFFFFF)) >> 9
            rows = (j * stride + jitter % stride).astype(np.int64)
            sc[q] += 1.0
            sc[q, rows] -= 2.0
    if kind == "mostlynan":
        sc[:, 40:] = np.nan
        sc[1, :] = np.nan
    want = OC.rank_full(sc)[:, :k]
    ids, vals = ops.topk(dev(sc), k, id_offset=7)
    np.testing.assert_array_equal(ids.cpu().numpy(), want + 7)
    np.testing.assert_array_equal(vals.cpu().numpy(), np.take_along_axis(sc, want, axis=1))
