"""Side legs of `bench.py --secondary` (single GPU): BASELINE.json's other single-GPU configurations and the rows f3 / f4 of
SURVEY.md section 8, timed BESIDE the headline on the same resident shard.  None of it is in the timed region of the bench
line; the default `bench.py` run does not execute it (round 6: headline + cpu_baseline + descriptors/s only).  The figures land
in the line as `sort_free_map_route` and `secondary_configs` (what tools/design_numbers.py reads).

    python bench.py --secondary [--steps K]
"""
import contextlib
import sys
import time

import numpy as np
import torch

N_ROXFORD = 4993
PEAK_HBM_GBS = 8000.0
SUM_ORDER_TOL = 2e-6        # bench.py: the summation-order bound two correct fp32 evaluations of one dot product are held to


def sort_free_leg(args, sharded, qvecs, sc, gnd, device, extra, NQ):
    """The headline's similarity kernel with the sort-free evaluation.  The two-stream leg of rounds 2-4 (ranking of batch k beside the similarity of batch k+1) is
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


def secondary_configs(args, ops, sharded, rows, qvecs, sc, rk, ws, gnd, device, extra, n_total, NQ, DIM):
    """configs[1] rOxford5k alone, the serving top-100, ONE evaluation on row-major descriptors, the labelled split-precision
    modes, configs[4]'s fp16 shard and its own 247tokyo1k shape, the float64 whitening-learning products, the CLAHE input
    conversion.  Overwrites ``sc`` (it ends up holding the fp16-shard scores).  Returns the dict of `secondary_configs`."""
    from mdir_amd.evaluate import compute_map_and_print, compute_map_and_print_from_scores
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
    return sec
