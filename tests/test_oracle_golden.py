"""The CPU oracle against outputs of the reference itself (tests/golden, made by
make_golden.py).  This is what pins the oracle; the GPU tests then compare the
HIP path with the oracle."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, sparse_map
from oracle import chain as OC
from oracle import oracle as O


def test_g1_pooling(golden):
    g = golden("g1_pool.npz")
    for c, h, w in [(2048, 24, 32), (2048, 17, 23), (512, 48, 64), (256, 7, 5)]:
        x = sparse_map(int(g[f"seed_c{c}_h{h}_w{w}"]), (1, c, h, w))
        for p in (3.0, 2.2, 1.0):
            np.testing.assert_allclose(O.gem(x, p)[0], g[f"gem_c{c}_h{h}_w{w}_p{p}"], rtol=5e-6, atol=1e-7)
        np.testing.assert_array_equal(O.mac(x)[0], g[f"mac_c{c}_h{h}_w{w}"])
        np.testing.assert_allclose(O.spoc(x)[0], g[f"spoc_c{c}_h{h}_w{w}"], rtol=5e-6)
    np.testing.assert_array_equal(sparse_map(103, (1, 256, 7, 5)), g["x_c256_h7_w5"])


def test_g2_l2n(golden):
    g = golden("g2_l2n.npz")
    y = O.l2n(g["x"])
    np.testing.assert_allclose(y, g["y"], rtol=1e-6, atol=1e-9)
    assert np.all(y[2] == 0) and not np.isnan(y).any()


def test_g3_forward_tail(golden):
    g = golden("g3_tail.npz")
    for p in (3.0, 2.92):
        np.testing.assert_allclose(O.forward_tail(g["feat"], p), g[f"out_plain_p{p}"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(O.forward_tail(g["feat"], p, whiten_w=g["w"], whiten_b=g["b"]),
                                   g[f"out_whiten_p{p}"], rtol=1e-5, atol=2e-7)


def test_g4_aggregate(golden):
    g = golden("g4_aggregate.npz")
    for msp in (1.0, 3.0, 2.92):
        np.testing.assert_allclose(O.ms_aggregate(g["vecs"], msp), g[f"agg_msp{msp}"], rtol=2e-6, atol=1e-8)


def test_g5_whiten(golden):
    g = golden("g5_whiten.npz")
    for dims in (None, 48):
        got = np.stack([O.whiten_wrapper(g["X"][:, i], g["m"], g["P"], dims) for i in range(g["X"].shape[1])], 1)
        np.testing.assert_allclose(got, g[f"wrapper_dims{dims}"], rtol=1e-5, atol=2e-7)
        np.testing.assert_allclose(O.whitenapply(g["X"].astype(np.float64), g["m"], g["P"], dims),
                                   g[f"whitenapply_f64_dims{dims}"], rtol=1e-12)
        f32 = O.whitenapply(g["X"], g["m"].astype(np.float32), g["P"].astype(np.float32), dims)
        assert f32.dtype == np.float32
        np.testing.assert_allclose(f32, g[f"whitenapply_f32_dims{dims}"], rtol=1e-5, atol=2e-7)


def test_g7_small_rankings(golden):
    g = golden("g7_ranking.npz")
    for name in "abc":
        sc = O.scores(g[f"{name}_vecs"], g[f"{name}_qvecs"])
        np.testing.assert_allclose(sc, g[f"{name}_scores"], rtol=0, atol=1e-6)
        assert float(g[f"{name}_mingap"]) > 1e-6
        np.testing.assert_array_equal(O.ranks(sc), g[f"{name}_ranks"])
        # the chain order the GPU commits to gives the same ranking on tie-free data
        ch = OC.scores_chain(g[f"{name}_vecs"], g[f"{name}_qvecs"])
        np.testing.assert_allclose(ch.T, g[f"{name}_scores"], rtol=0, atol=1e-6)
        np.testing.assert_array_equal(OC.rank_full(ch).T, g[f"{name}_ranks"])


def test_g7_ties(golden):
    g = golden("g7_ranking.npz")
    sc = O.scores(g["tie_vecs"], g["tie_qvecs"])
    np.testing.assert_array_equal(sc, g["tie_scores"])  # exactly representable sums
    mine, ref = O.ranks(sc), g["tie_ranks_numpy_default"]
    # same ids inside every run of equal scores; the build orders a run by ascending id
    for q in range(sc.shape[1]):
        col = sc[:, q]
        np.testing.assert_array_equal(col[mine[:, q]], col[ref[:, q]])
        srt = col[mine[:, q]]
        for v in np.unique(srt):
            run = np.nonzero(srt == v)[0]
            assert set(mine[run, q]) == set(ref[run, q])
            assert np.all(np.diff(mine[run, q]) > 0)
    np.testing.assert_array_equal(OC.rank_full(np.ascontiguousarray(sc.T)).T, mine)


def test_g7_roxford_shape(golden):
    g = golden("g7_roxford_shape.npz")
    vecs, qvecs, qid = O.synth_ranking_problem(4993, 70, 2048, seed=0)
    assert float(g["vecs_checksum"]) == float(vecs.astype(np.float64).sum())
    np.testing.assert_array_equal(qid, g["qid"])
    sc = O.scores(vecs, qvecs)
    np.testing.assert_allclose(sc[::97], g["scores_rows_0_4992_step_97"], rtol=0, atol=1e-6)
    rk = O.ranks(sc)
    # the top-100 are separated by more than fp32 summation noise -> identical ids
    assert float(g["top100_mingap"]) > 1e-8
    same = rk[:100] == g["top100"]
    assert same.mean() > 0.999
    np.testing.assert_array_equal(rk[0], qid)  # every query finds its source row first
    avg, per = O.compute_map_and_print("roxford5k", rk, O.synth_gnd(70, 4993, seed=1))
    for lvl in ("easy", "medium", "hard"):
        np.testing.assert_allclose(avg["map_" + lvl], float(g["map_" + lvl]), rtol=0, atol=1e-6)
        np.testing.assert_allclose(per["ap_" + lvl], g["ap_" + lvl], rtol=0, atol=1e-5)


def test_g8_compute_map(golden):
    g = golden("g8_map.npz")
    rk = g["ranks"].astype(np.int64)
    gnd = json.loads(bytes(g["gnd_json"]).decode())
    avg, per = O.compute_map_and_print("roxford5k", rk, gnd)
    for lvl in ("easy", "medium", "hard"):
        assert avg["map_" + lvl] == float(g["rox_map_" + lvl])
        np.testing.assert_array_equal(per["ap_" + lvl], g["rox_ap_" + lvl])
    gnd_m = O.protocol_gnd(gnd, "medium")
    m, aps, pr, prs = O.compute_map(rk, gnd_m, [1, 5, 10])
    assert m == float(g["medium_map"])
    np.testing.assert_array_equal(aps, g["medium_aps"])
    np.testing.assert_array_equal(pr, g["medium_pr"])
    np.testing.assert_array_equal(prs, g["medium_prs"])
    assert np.isnan(aps[4]) and np.isnan(prs[4]).all()
    old = [{"ok": x["easy"] + x["hard"], "junk": x["junk"]} for x in gnd]
    avg, per = O.compute_map_and_print("247tokyo1k", rk, old)
    assert avg["map"] == float(g["old_map"])
    np.testing.assert_array_equal(per["ap"], g["old_ap"])
    m, aps, _, _ = O.compute_map(rk, [{"ok": x["ok"]} for x in old])
    assert m == float(g["nojunkkey_map"])
    np.testing.assert_array_equal(aps, g["nojunkkey_aps"])
    assert bool(g["other_dataset_returns_none"]) and O.compute_map_and_print("oxford5k", rk, gnd) is None
    assert O.compute_ap([0], 1) == float(g["ap_r0_n1"]) == 1.0
    assert O.compute_ap([1], 1) == float(g["ap_r1_n1"]) == 0.25
    assert O.compute_ap([0, 2], 2) == float(g["ap_r02_n2"])
    assert O.compute_ap([], 3) == float(g["ap_empty_n3"]) == 0
    assert O.compute_ap([0, 4, 9], 5) == float(g["ap_r0_4_9_n5"])


def test_rank_of_matches_full_ranking():
    rng = np.random.default_rng(5)
    sc = rng.standard_normal((400, 3)).astype(np.float32)
    sc[10:20, 1] = sc[5, 1]          # a run of ties
    sc[30, 2] = np.nan
    rk = O.ranks(sc[:, :2])
    ids = np.array([5, 10, 19, 0, 399, 77])
    for q in range(2):
        inv = np.empty(400, dtype=np.int64)
        inv[rk[:, q]] = np.arange(400)
        np.testing.assert_array_equal(O.rank_of(sc[:, q], ids), inv[ids])
        np.testing.assert_array_equal(OC.rank_of(sc[:, q], ids), inv[ids])
    # C ranking: NaN goes last, -0 == +0
    crk = OC.rank_full(np.ascontiguousarray(sc.T))
    assert crk[2, -1] == 30
    assert OC.desc_key(-0.0) == OC.desc_key(0.0)
    for q in range(2):
        np.testing.assert_array_equal(crk[q], rk[:, q])


def _fma32(a, b, c):
    """fl32(a*b + c) with ONE rounding, computed exactly: rational arithmetic, then the nearest float32
    (numpy rounds a python float to float32 correctly, and the exact rational is turned into the nearest
    double first only when that double is already exact enough -- so round directly from the rational)."""
    from fractions import Fraction
    exact = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    if exact == 0:
        return np.float32(0.0)
    # nearest float32 of a rational: scale to an integer significand of 24 bits, round half to even
    sign = -1 if exact < 0 else 1
    mag = abs(exact)
    e = mag.numerator.bit_length() - mag.denominator.bit_length()
    if Fraction(2) ** e > mag:
        e -= 1                                                  # 2^e <= mag < 2^(e+1)
    e = max(e, -126)                                            # subnormal range keeps the exponent of 2^-126
    scaled = mag / Fraction(2) ** (e - 23)                      # significand in [2^23, 2^24) (or below, if subnormal)
    n, rem = divmod(scaled.numerator, scaled.denominator)
    twice = 2 * rem
    if twice > scaled.denominator or (twice == scaled.denominator and n % 2 == 1):
        n += 1
    return np.float32(sign * float(Fraction(n) * Fraction(2) ** (e - 23)))


def test_chain_is_sequential_fma():
    """oracle/chain.c really is the k-ascending chain of fused multiply-adds, one rounding per term.  Checked against an
    EXACT fma (rational arithmetic, correctly rounded to float32) -- a float64 emulation can double-round."""
    rng = np.random.default_rng(6)
    d, n, q = 37, 9, 3
    vecs = rng.standard_normal((d, n)).astype(np.float32)
    qv = rng.standard_normal((d, q)).astype(np.float32)
    vecs[:, 0] *= np.float32(1e-20)                              # products in the subnormal range too
    got = OC.scores_chain(vecs, qv)
    for j in range(q):
        for i in range(n):
            acc = np.float32(0)
            for k in range(d):
                acc = _fma32(qv[k, j], vecs[k, i], acc)
            assert got[j, i] == acc, (j, i, got[j, i], acc)
    # the rounding helper itself: halfway cases go to even, and it agrees with numpy where no tie is involved
    assert _fma32(np.float32(1.0), np.float32(1.0), np.float32(2.0 ** -24)) == np.float32(1.0)
    assert _fma32(np.float32(1.0), np.float32(1.0 + 2.0 ** -23), np.float32(2.0 ** -24)) == np.float32(1.0 + 2.0 ** -22)
    a = rng.standard_normal((5, d)).astype(np.float32)
    b = rng.standard_normal((4, d)).astype(np.float32)
    np.testing.assert_array_equal(OC.gemm_nt_chain(a, b), OC.scores_chain(b.T.copy(), a.T.copy()))


def test_g11_nanmean_metric():
    meta = json.load(open(os.path.join(GOLDEN, "g11_scenario.json")))["metadata"]
    aps = [0.5, float("nan"), 0.25, 1.0]
    assert O.nanmean_metric(aps) == meta["roxford5k/validation/score:ap_medium_avg.4"][0]


# ------------------------------------------------------------------ rows marked "next" (f1-f3) and the loader (a1)

def _rows_up_to_sign(a, b):
    """Rows of eigenvector-derived matrices are defined up to sign."""
    sign = np.sign(np.sum(a * b, axis=1, keepdims=True))
    return a * sign


def test_g12_whitening_learning(golden):
    g = golden("g12_whitenlearn.npz")
    m, P = O.whitenlearn(g["X"], g["qidxs"], g["pidxs"])
    np.testing.assert_allclose(m, g["m_lw"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(_rows_up_to_sign(P, g["P_lw"]), g["P_lw"], rtol=1e-9, atol=1e-10)
    m2, P2 = O.pcawhitenlearn(g["X"])
    np.testing.assert_allclose(m2, g["m_pca"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(_rows_up_to_sign(np.real(P2), g["P_pca"]), g["P_pca"], rtol=1e-8, atol=1e-9)
    _, P3 = O.pcawhitenlearn(g["X"], shrink=8)
    np.testing.assert_allclose(_rows_up_to_sign(np.real(P3), g["P_pca_shrink8"]), g["P_pca_shrink8"], rtol=1e-8, atol=1e-9)
    np.testing.assert_array_equal(O.cholesky_bumped(g["S_singular"]), g["L_singular"])      # needed the diagonal bump
    np.testing.assert_array_equal(O.cholesky_bumped(g["S_pd"]), g["L_pd"])


def test_g13_hard_negative_selection(golden):
    g = golden("g13_mining.npz")
    nidxs, ndist = O.hard_negatives(g["qvecs"], g["poolvecs"], g["idxs2images"], g["clusters"].tolist(), g["qidxs"].tolist(),
                                    int(g["nnum"]))
    assert nidxs == g["nidxs"].tolist()
    np.testing.assert_allclose(ndist, g["ndist"], rtol=1e-5)


def test_g14_embedding_output(golden):
    g = golden("g14_embedding_output.npz")
    got = O.embedding_output(4, [g["vec"][0], None, g["vec"][2], g["vec"][3]])
    assert got.dtype == np.float64 == np.dtype(str(g["result_dtype"]))
    np.testing.assert_array_equal(got, g["result"])
    assert O.embedding_output(4, []) == [] and g["empty_second"][0] == "[]"


def test_g15_image_loader(golden, tmp_path):
    g = golden("g15_loader.npz")
    for name in ("landscape", "portrait", "small"):
        (tmp_path / (name + ".png")).write_bytes(g["file_" + name].tobytes())
    ncases = sum(1 for k in g.files if k.endswith("_spec"))
    assert ncases == 6
    for ci in range(ncases):
        name, imsize, bbx = eval(str(g["case%d_spec" % ci][0]))
        np.testing.assert_array_equal(O.load_image(str(tmp_path / (name + ".png")), imsize, bbx), g["case%d_out" % ci])


def test_thumbnail_restatement_is_pillow_and_g15(golden, tmp_path):
    """The restated thumbnail (size rule, LANCZOS taps in Pillow's fixed point, two uint8 passes) equals Pillow's
    ``Image.thumbnail`` pixel for pixel -- Pillow is what the reference calls (datahelpers.py:48-50) -- on random
    images, and reproduces golden G15 (outputs of the reference's loader) when it replaces Pillow's resize there."""
    from PIL import Image
    rng = np.random.default_rng(4)
    for w, h, imsize in [(221, 150, 64), (97, 203, 64), (640, 480, 362), (1025, 700, 1024), (333, 1000, 500), (800, 600, 1024),
                         (1023, 4, 512), (9, 890, 300)]:
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(arr)
        im.thumbnail((imsize, imsize), Image.LANCZOS)
        np.testing.assert_array_equal(O.thumbnail_u8(arr, imsize), np.asarray(im))
        assert O.thumbnail_size(w, h, imsize) in (None, im.size)
    for _ in range(300):
        w, h, imsize = int(rng.integers(1, 3000)), int(rng.integers(1, 3000)), int(rng.choice([64, 362, 1024]))
        im = Image.new("L", (w, h))
        im.thumbnail((imsize, imsize), Image.LANCZOS)
        assert (O.thumbnail_size(w, h, imsize) or (w, h)) == im.size
    g = golden("g15_loader.npz")
    for name in ("landscape", "portrait", "small"):
        (tmp_path / (name + ".png")).write_bytes(g["file_" + name].tobytes())
    for ci in range(6):
        name, imsize, bbx = eval(str(g["case%d_spec" % ci][0]))
        raw = O.load_image(str(tmp_path / (name + ".png")), None, bbx)
        got = raw if imsize is None else O.thumbnail_u8(raw, imsize)
        np.testing.assert_array_equal(got, g["case%d_out" % ci])


def _jpeg_cases():
    import io
    from PIL import Image
    rng = np.random.default_rng(21)

    def picture(w, h, kind):
        if kind == "sat":
            a = rng.integers(0, 2, (h, w, 3)) * 255
        elif kind == "noise":
            a = rng.integers(0, 256, (h, w, 3))
        elif kind == "photo":
            low = rng.integers(0, 255, (h // 8 + 1, w // 8 + 1, 3)).astype(np.float32)
            a = np.clip(np.kron(low, np.ones((8, 8, 1), np.float32))[:h, :w] + rng.normal(0, 15, (h, w, 3)), 0, 255)
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            a = np.stack([xx * 255 // max(w - 1, 1), yy * 255 // max(h - 1, 1), (xx * 7 + yy * 13) % 256], axis=2)
        return Image.fromarray(a.astype(np.uint8))

    out = []
    for w, h, kind, q, sub, kw in [(64, 48, "photo", 90, 0, {}), (65, 47, "photo", 75, 2, {}), (131, 77, "photo", 85, 1, {}),
                                   (33, 50, "photo", 95, 2, {"optimize": True}), (17, 19, "noise", 80, 2, {}), (16, 16, "grad", 30, 1, {}),
                                   (97, 61, "sat", 1, 2, {}), (97, 61, "sat", 100, 0, {}), (97, 61, "noise", 10, 1, {}), (97, 61, "grad", 100, 2, {}),
                                   (300, 200, "noise", 85, 2, {"restart_marker_blocks": 7}), (300, 200, "grad", 85, 1, {"restart_marker_rows": 1}),
                                   (640, 480, "photo", 90, 2, {})]:
        buf = io.BytesIO()
        picture(w, h, kind).save(buf, format="JPEG", quality=q, subsampling=sub, **kw)
        out.append(("%dx%d %s q%d sub%d %s" % (w, h, kind, q, sub, kw), buf.getvalue()))
    buf = io.BytesIO()
    picture(200, 120, "photo").convert("L").save(buf, format="JPEG", quality=50)
    out.append(("grey", buf.getvalue()))
    for w, h, kind, q, sub in [(97, 61, "photo", 92, 2), (256, 160, "photo", 50, 1), (33, 47, "noise", 75, 0), (97, 61, "sat", 5, 2),
                               (131, 77, "grad", 100, 2)]:
        buf = io.BytesIO()
        picture(w, h, kind).save(buf, format="JPEG", quality=q, subsampling=sub, progressive=True)
        out.append(("progressive %dx%d %s q%d sub%d" % (w, h, kind, q, sub), buf.getvalue()))
    buf = io.BytesIO()
    picture(200, 120, "grad").convert("L").save(buf, format="JPEG", quality=70, progressive=True)
    out.append(("progressive grey", buf.getvalue()))
    return out, picture


def test_jpeg_restatement_is_pillow():
    """libjpeg's decompression restated (host: the library's entropy decoder, the serial half of the product's JPEG path;
    oracle: dequantisation, islow IDCT, fancy upsampling, YCbCr -> RGB) equals Pillow's decode of the same file -- Pillow is
    what the reference calls (datahelpers.py:24-31) -- pixel for pixel: 4:4:4 / 4:2:2 / 4:2:0 / grey, odd sizes, qualities
    1..100, saturated pictures, optimised Huffman tables, restart markers, progressive files (spectral selection + successive
    approximation); CMYK, non-JPEG, truncated and corrupt files are declined (they stay with Pillow)."""
    import ctypes
    import io
    from PIL import Image
    from mdir_amd import _lib
    lib = _lib.lib()
    cases, picture = _jpeg_cases()

    def decode(data):
        buf = np.frombuffer(data, dtype=np.uint8)
        info = _lib.JpegInfo()
        assert lib.mdx_jpeg_probe(buf.ctypes.data, buf.size, ctypes.byref(info)) == 0
        if not info.supported:
            return None
        coef, quant = np.empty((info.nblocks, 64), dtype=np.int16), np.empty((3, 64), dtype=np.uint16)
        assert lib.mdx_jpeg_coefficients(buf.ctypes.data, buf.size, coef.ctypes.data, info.nblocks, quant.ctypes.data) == 0
        return O.jpeg_pixels(coef, quant, {k: (list(getattr(info, k)) if k.endswith(("samp", "_w", "_h", "offset")) else getattr(info, k))
                                           for k in ("width", "height", "ncomp", "hsamp", "vsamp", "blocks_w", "blocks_h", "block_offset")})

    for name, data in cases:
        want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        got = decode(data)
        assert got is not None, name
        np.testing.assert_array_equal(got, want, err_msg=name)
    buf = io.BytesIO()
    picture(64, 64, "noise").convert("CMYK").save(buf, format="JPEG")
    assert decode(buf.getvalue()) is None
    buf = io.BytesIO()
    picture(64, 64, "noise").save(buf, format="PNG")
    assert decode(buf.getvalue()) is None

    def declined(data):
        b = np.frombuffer(data, dtype=np.uint8)
        info = _lib.JpegInfo()
        lib.mdx_jpeg_probe(b.ctypes.data, b.size, ctypes.byref(info))
        if not info.supported:
            return True
        coef, quant = np.empty((info.nblocks, 64), dtype=np.int16), np.empty((3, 64), dtype=np.uint16)
        return lib.mdx_jpeg_coefficients(b.ctypes.data, b.size, coef.ctypes.data, info.nblocks, quant.ctypes.data) != 0

    whole = cases[12][1]                                                  # 640x480
    for cut in (len(whole) // 2, len(whole) - 3, 700):
        assert declined(whole[:cut])                                      # data runs out inside the scan
    # a progressive file that ends after a complete scan (no further SOS, no EOI): Pillow calls it truncated, so does the
    # host stage; same for a file that only lacks its end-of-image marker
    prog = [d for name, d in cases if "prog" in name][0]
    sos = [i for i in range(len(prog) - 1) if prog[i] == 0xFF and prog[i + 1] == 0xDA]
    assert len(sos) >= 4
    assert declined(prog[:sos[3]])
    assert whole[-2:] == b"\xff\xd9" and declined(whole[:-2])
    # a quantisation table redefined between two scans (libjpeg latches tables per component at its first scan)
    dqt = prog.index(b"\xff\xdb")
    seg = prog[dqt:dqt + 2 + int.from_bytes(prog[dqt + 2:dqt + 4], "big")]
    assert declined(prog[:sos[1]] + seg + prog[sos[1]:])
    assert not declined(prog)
    broken = bytearray(whole)
    for i in range(1000, len(broken), 997):
        broken[i] ^= 0x5A
    declined(bytes(broken))                                               # anything but a crash


def test_jpeg_host_stage_on_damaged_files():
    """Damaged files (random bytes overwritten, tails cut off) never crash the host entropy stage; most are declined (they
    go to Pillow, whose tolerance the reference relies on: LOAD_TRUNCATED_IMAGES); those it still decodes must almost
    always be what Pillow makes of the same bytes (a few damaged progressive files are handled differently by libjpeg)."""
    import ctypes
    import io
    import warnings
    from PIL import Image, ImageFile
    from mdir_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(33)
    old = ImageFile.LOAD_TRUNCATED_IMAGES
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    same = differ = declined = 0
    try:
        for prog in (False, True):
            for sub in (0, 2):
                low = rng.integers(0, 255, (12, 17, 3)).astype(np.float32)
                img = np.clip(np.kron(low, np.ones((8, 8, 1), np.float32))[:90, :130] + rng.normal(0, 12, (90, 130, 3)), 0, 255).astype(np.uint8)
                buf = io.BytesIO()
                Image.fromarray(img).save(buf, format="JPEG", quality=80, subsampling=sub, progressive=prog)
                base = bytearray(buf.getvalue())
                for trial in range(120):
                    d = bytearray(base)
                    for _ in range(int(rng.integers(1, 4))):
                        d[int(rng.integers(2, len(d)))] = int(rng.integers(0, 256))
                    if trial % 5 == 0:
                        d = d[:int(rng.integers(20, len(d)))]
                    data = bytes(d)
                    b = np.frombuffer(data, dtype=np.uint8)
                    info = _lib.JpegInfo()
                    lib.mdx_jpeg_probe(b.ctypes.data, b.size, ctypes.byref(info))
                    if not info.supported or info.nblocks > 100000:
                        declined += 1
                        continue
                    coef, quant = np.empty((info.nblocks, 64), dtype=np.int16), np.empty((3, 64), dtype=np.uint16)
                    if lib.mdx_jpeg_coefficients(b.ctypes.data, b.size, coef.ctypes.data, info.nblocks, quant.ctypes.data) != 0:
                        declined += 1
                        continue
                    meta = {k: (list(getattr(info, k)) if k.endswith(("samp", "_w", "_h", "offset")) else getattr(info, k))
                            for k in ("width", "height", "ncomp", "hsamp", "vsamp", "blocks_w", "blocks_h", "block_offset")}
                    got = O.jpeg_pixels(coef, quant, meta)
                    try:
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            im = Image.open(io.BytesIO(data))
                            if im.size != (info.width, info.height) or im.mode not in ("RGB", "L"):
                                continue                          # the loader compares these too and takes the host route
                            want = np.asarray(im.convert("RGB"))
                    except Exception:
                        continue                                  # ... as it does when Pillow refuses the file
                    if np.array_equal(got, want):
                        same += 1
                    else:
                        differ += 1
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = old
    assert declined > 100 and same > 50 and differ <= max(2, same // 50), (same, differ, declined)


def test_clahe_restatement_properties():
    """The CLAHE restatement (OpenCV's published algorithm; PARITY UNPINNED against OpenCV itself, which is absent here) at
    least has the algorithm's defining properties: Lab round trip, LUTs monotone with the clip limit honoured, identity
    blending on a constant plane, OpenCV's padding rule, output in range."""
    rng = np.random.default_rng(3)
    x = rng.random((31, 45, 3), dtype=np.float32)
    np.testing.assert_allclose(O.lab_to_rgb(O.rgb_to_lab(x)), x, atol=5e-5)
    lab = O.rgb_to_lab(np.array([[[1, 1, 1], [0, 0, 0], [1, 0, 0]]], dtype=np.float32))
    np.testing.assert_allclose(lab[0, 0], [100, 0, 0], atol=2e-3)                 # white
    np.testing.assert_allclose(lab[0, 1], [0, 0, 0], atol=1e-5)                   # black
    np.testing.assert_allclose(lab[0, 2], [53.24, 80.09, 67.20], atol=2e-2)       # sRGB red (published Lab value)
    l8 = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    luts, tile = O.clahe_luts(l8, 4, (8, 8))
    assert tile == (8, 12) and luts.shape == (8, 8, 256)
    assert (np.diff(luts.astype(np.int32), axis=2) >= 0).all() and (luts[..., -1] == 255).all()
    # slope bound: a bin holds at most clip + spread pixels -> LUT steps of at most (clip + 2) * 255 / area
    clip = max(int(4 * 96 / 256), 1)
    assert np.diff(luts.astype(np.int32), axis=2).max() <= np.ceil((clip + 2) * 255 / 96) + 1
    # a side that is not a multiple of the grid pads BOTH sides (OpenCV's rule): 60 x 96 -> tiles of (64/8, 104/8)
    assert O.clahe_luts(rng.integers(0, 256, (60, 96), dtype=np.uint8), 4, (8, 8))[1] == (8, 13)
    const = np.full((64, 64), 77, dtype=np.uint8)
    lc, tc = O.clahe_luts(const, 4, (8, 8))
    assert len(np.unique(O.clahe_apply(const, lc, tc))) == 1                       # every tile has the same LUT
    out, _ = O.apply_clahe_rgb(rng.integers(0, 256, (37, 53, 3), dtype=np.uint8))
    assert out.dtype == np.float32 and out.min() >= 0 and out.max() <= 1


def test_split_precision_restatements():
    """The oracle's restatements of the two labelled split-precision modes (MDX_F32_SPLIT3: three bf16 pieces; MDX_F32_SPLIT2:
    block floating point, two fp16 pieces) -- what the GPU tests compare the kernels with: the pieces are representable in
    their formats, reconstruct the operand to the stated bounds, and the scores sit inside the modes' error bounds against
    the float64 product."""
    rng = np.random.default_rng(21)
    x = (rng.standard_normal(20000) * np.exp(rng.uniform(-30, 30, 20000))).astype(np.float32)
    x[:3] = [0.0, -0.0, np.float32(1.17549435e-38)]
    h, m, l = O.split3_bf16(x)
    for piece in (h, m, l):                                     # a bf16 value: the low 16 bits of the fp32 pattern are zero
        assert (piece.view(np.uint32) & 0xFFFF == 0).all()
    rec = h.astype(np.float64) + m + l
    ok = np.abs(x) > 1e-30                                      # (pieces of denormal-sized residuals flush; not the path's data)
    assert (np.abs(x[ok] - rec[ok]) <= np.abs(x[ok]) * 2.0 ** -24).all()
    # bf16 rounding is to nearest even: against the definition on a few hand cases
    hand = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7, 1.0 + 3 * 2.0 ** -8, -3.1415927], np.float32)
    np.testing.assert_array_equal(O.bf16_round(hand), np.array([1.0, 1.0, 1.0 + 2.0 ** -7, 1.0 + 2.0 ** -6, -3.140625], np.float32))

    # fp16 toward zero: equal to numpy's float16 wherever that conversion is exact or rounds down in magnitude
    v = np.concatenate([rng.standard_normal(5000) * 100, rng.standard_normal(5000) * 1e-5, [0.0, 65504.0, 6.1035e-05, 5.96e-08]])
    t = O._fp16_toward_zero(v)
    assert (np.abs(t) <= np.abs(v)).all() and (np.sign(t) * np.sign(v) >= 0).all()
    assert (t.astype(np.float16).astype(np.float64) == t).all()             # representable in fp16
    ulp = np.where(np.abs(v) >= 2.0 ** -14, np.exp2(np.floor(np.log2(np.maximum(np.abs(v), 1e-300))) - 10), 2.0 ** -24)
    assert (np.abs(v - t) < ulp).all()
    for a in (np.array([0.02, -0.7]), np.array([1e-20, 3e-21]), np.array([5e20]), np.zeros(4)):
        s = O.split2_scale(a.astype(np.float32))
        top = np.abs(a).max()
        assert s == 1.0 if top == 0 else (np.log2(s) == np.round(np.log2(s)) and 2.0 ** 13 <= top * s < 2.0 ** 14)

    vecs, qvecs, _ = O.synth_ranking_problem(3000, 40, 512, seed=5)
    exact = vecs.astype(np.float64).T @ qvecs.astype(np.float64)
    assert np.abs(O.scores_split3(vecs, qvecs) - exact).max() < 6e-8        # 2^-23 of the products + the final fp32 rounding
    # split2 truncates toward zero twice per operand: every piece sum falls short of its operand by up to 2^-20 of it, so a
    # score falls short by up to 2^-19 of itself -- a bias proportional to the score (it shrinks all scores alike: neutral
    # for a ranking) -- plus a far smaller random part
    s2 = O.scores_split2(vecs, qvecs)
    assert (np.abs(s2 - exact) <= 2.0 ** -19 * np.abs(exact) + 5e-8).all()
    assert np.abs(s2 - exact).max() < 1e-6
    # block exponents are exact: other powers of two in, the same scores times that power out
    np.testing.assert_array_equal(O.scores_split2(vecs * np.float32(2.0 ** 9), qvecs * np.float32(2.0 ** -20)),
                                  O.scores_split2(vecs, qvecs) * np.float32(2.0 ** -11))


def test_g17_rmac(golden):
    """R-MAC (functional.py:26-72): the oracle's float32 region grid and region-ordered sum against the reference's outputs."""
    g = golden("g17_rmac.npz")
    for c, h, w, b in [(2048, 24, 32, 1), (512, 48, 64, 1), (64, 17, 23, 2), (256, 7, 5, 1), (16, 3, 40, 2), (8, 12, 12, 1), (4, 2, 2, 1)]:
        x = sparse_map(int(g["seed_c%d_h%d_w%d_b%d" % (c, h, w, b)]), (b, c, h, w))
        for L in (3, 2):
            np.testing.assert_allclose(O.rmac(x, L=L), g["rmac_c%d_h%d_w%d_b%d_L%d" % (c, h, w, b, L)], rtol=2e-6, atol=2e-6)
    assert len(O.rmac_regions(24, 32, 3)) == 20 and len(O.rmac_regions(12, 12, 3)) == 14 and len(O.rmac_regions(3, 40, 3)) == 50
    assert O.rmac_regions(1, 5, 3) == O.rmac_regions(1, 5, 1)                  # windows of size 0 (levels 2, 3 of a 1-pixel side) are skipped


def test_g18_rpool(golden):
    """Regional pooling (functional.py:75-121, pooling.py:62-95): the oracle against the reference's Rpool outputs."""
    g = golden("g18_rpool.npz")
    pools = {"gem": lambda a: O.gem(a, 2.5, 1e-6), "mac": O.mac, "spoc": O.spoc}
    for c, h, w in [(64, 24, 32), (32, 17, 23), (16, 7, 5), (8, 12, 12)]:
        x = sparse_map(int(g["seed_c%d_h%d_w%d" % (c, h, w)]), (2, c, h, w))
        for name, fn in pools.items():
            for tag, wb in (("plain", (None, None)), ("whiten", (g["weight_c%d" % c], g["bias_c%d" % c]))):
                np.testing.assert_allclose(O.rpool(x, fn, wb[0], wb[1]), g["agg_%s_%s_c%d_h%d_w%d" % (name, tag, c, h, w)], rtol=1e-5, atol=2e-6)
                np.testing.assert_allclose(O.rpool(x, fn, wb[0], wb[1], aggregate=False), g["reg_%s_%s_c%d_h%d_w%d" % (name, tag, c, h, w)],
                                           rtol=1e-5, atol=2e-6)
