#!/usr/bin/env python3
"""Regenerate the "current numbers" table of DESIGN.md from the committed evidence of a round:
   profiles/<tag>_bench_line.json   the one JSON line of `python bench.py` on an MI355X
   profiles/<tag>_traffic.json      HBM bytes per launch from the PMC passes (tools/summarize_profile.py)
Everything between `<!-- NUMBERS:BEGIN -->` and `<!-- NUMBERS:END -->` in DESIGN.md is replaced.
    python tools/design_numbers.py r06"""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = json.load(open(os.path.join(root, "profiles", tag + "_bench_line.json")))
t = json.load(open(os.path.join(root, "profiles", tag + "_traffic.json")))
r, rr, sec = b["roofline"], b.get("roofline_rank", {}), b.get("secondary_configs", {})
cpu, ds = b.get("cpu_baseline", {}), b.get("descriptors_per_s", {})
# round 6: the side legs are a second run (`bench.py --secondary`), committed as <tag>_secondary_line.json
sec_path = os.path.join(root, "profiles", tag + "_secondary_line.json")
if not sec and os.path.exists(sec_path):
    b2 = json.load(open(sec_path))
    sec = b2.get("secondary_configs", {})
    b.setdefault("sort_free_map_route", b2.get("sort_free_map_route"))
    for mode in ("split3", "split2"):          # (the second run's own exact-chain time stands beside its split times)
        if mode in sec and not sec[mode].get("exact_chain_scores_ms"):
            sec[mode]["exact_chain_scores_ms"] = b2["roofline"]["kernel_ms"]
arb = (b.get("cpu_path_parity") or {}).get("f64_arbiter") or {}


def g(d, *keys, default=None):
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def f(x, fmt="%.3f"):
    return "-" if x is None else fmt % x


rows = [
    ("headline: queries/s, exact full ranking (configs[2]: N = 1 004 993, Q = 70, D = 2048 fp32)",
     "%s queries/s = %s ms per 70-query batch" % (f(b["value"], "%.0f"), f(b["ms_per_step"], "%.3f")), "`value`, `ms_per_step`"),
    ("similarity kernel (exact fp32 chain), HIP events in the timed region",
     "%s ms = %s TFLOP/s = **%s of the fp32 MFMA peak** (157.3); %s of the peak at the profiled clock (%s GHz, pipe busy %s)"
     % (f(r["kernel_ms"]), f(r["achieved"], "%.1f"), f(r["frac"]), f(r.get("frac_of_peak_at_profiled_clock")),
        f(r.get("profiled_sustained_clock_ghz"), "%.2f"), f(r.get("profiled_mfma_pipe_busy"), "%.2f")), "`roofline`"),
    ("  the same kernel in the committed rocprofv3 kernel trace (same command under the profiler)",
     "%s us average = **%s of the peak**" % (f(r.get("rocprof_kernel_avg_us"), "%.1f"), f(r.get("frac_rocprof"))), "`roofline.frac_rocprof`"),
    ("  its HBM traffic by PMC / algorithmic bytes", "%s GB / %s GB = %sx" % (f(r["traffic"] / 1e9 if r.get("traffic") else None),
                                                                             f(r["algorithmic_bytes"] / 1e9),
                                                                             f(r["traffic"] / r["algorithmic_bytes"] if r.get("traffic") else None, "%.2f")),
     "`roofline.traffic` (`profiles/%s_traffic.json`)" % tag),
    ("exact full ranking (4-pass LSD radix sort), HIP events", "%s ms; algorithmic 0.844 GB -> %s GB/s = %s of 8 TB/s; real traffic %s GB (%sx) at %s TB/s"
     % (f(rr.get("kernel_ms")), f(rr.get("achieved"), "%.0f"), f(rr.get("frac")), f(rr["traffic"] / 1e9 if rr.get("traffic") else None, "%.2f"),
        f(rr.get("traffic_over_algorithmic"), "%.2f"), f(rr["hbm_GBps_at_real_traffic"] / 1e3 if rr.get("hbm_GBps_at_real_traffic") else None, "%.2f")),
     "`roofline_rank`"),
    ("step time not inside the two kernel families (launch gaps)", "%s ms" % f(b.get("step_ms_minus_kernels"), "%.4f"), "`step_ms_minus_kernels`"),
    ("mAP-medium (synthetic rOxford-shaped labels): GPU / numpy CPU path", "%s / %s" % (f(b.get("map_medium"), "%.10f"), f(b.get("map_medium_cpu"), "%.10f")),
     "`map_medium`, `map_medium_cpu`, `cpu_path_parity`"),
    ("  float64 arbiter: ranking slots where GPU and CPU differ; of those the GPU / the CPU / neither names float64's row; mAP under the float64 order",
     "%s of %s; %s / %s / %s; %s" % (arb.get("slots_where_gpu_and_cpu_differ"), arb.get("of_slots"), arb.get("gpu_order_agrees_with_f64"),
                                    arb.get("cpu_order_agrees_with_f64"), arb.get("neither_agrees_with_f64"), f(arb.get("map_medium_f64_order"), "%.10f")),
     "`cpu_path_parity.f64_arbiter`"),
    ("  largest |score - float64 score|: GPU chain / BLAS; largest float64 gap between two rows a path orders the other way: GPU / CPU",
     "%s / %s; %s / %s (bound 2e-6)" % (f(arb.get("gpu_max_abs_score_error_vs_f64"), "%.2e"), f(arb.get("cpu_max_abs_score_error_vs_f64"), "%.2e"),
                                         f(arb.get("gpu_max_f64_gap_between_misordered_rows"), "%.2e"), f(arb.get("cpu_max_f64_gap_between_misordered_rows"), "%.2e")),
     "`cpu_path_parity.f64_arbiter`"),
    ("CPU reference beside it (np.dot + np.argsort, %s host cores)" % cpu.get("cores"), "%s queries/s (%s with the BLAS pool at 3 threads)"
     % (f(cpu.get("value"), "%.1f"), f(cpu.get("value_blas_3_threads"), "%.1f")), "`cpu_baseline`"),
    ("sort-free evaluation route (similarity + rank counting, same mAP)", "%s queries/s" % f(g(b, "sort_free_map_route", "value"), "%.0f"), "`sort_free_map_route`"),
    ("spread over the timed steps of this run (min / median / max): similarity ms; ranking ms; queries/s",
     "%s / %s / %s; %s / %s / %s; %s / %s / %s" % tuple(f(g(b, "spread_over_timed_steps", k, m), fmt) for k, fmt in (("kernel_ms", "%.3f"), ("rank_ms", "%.3f"), ("value", "%.0f"))
                                                      for m in ("min", "median", "max")), "`spread_over_timed_steps`"),
    ("LABELLED split-precision mode `MDX_F32_SPLIT3` on the same shard", "%s ms (exact chain %s) = %s GB/s of algorithmic bytes = %s of 8 TB/s; max abs(diff) %s; top-100 slot agreement %s; with the fp32 ranking %s queries/s"
     % (f(g(sec, "split3", "scores_ms")), f(g(sec, "split3", "exact_chain_scores_ms")), f(g(sec, "split3", "roofline", "achieved"), "%.0f"),
        f(g(sec, "split3", "roofline", "frac")), f(g(sec, "split3", "max_abs_diff_vs_exact_chain"), "%.1e"),
        f(g(sec, "split3", "top100_slot_agreement_with_exact"), "%.4f"), f(g(sec, "split3", "queries_per_s_with_the_fp32_ranking"), "%.0f")),
     "`secondary_configs.split3`, `profiles/%s_split3.md`" % tag),
    ("LABELLED block-floating mode `MDX_F32_SPLIT2` (two fp16 pieces, three products)", "%s ms = %s GB/s of algorithmic bytes = %s of 8 TB/s; max abs(diff) %s; top-100 slot agreement %s; with the fp32 ranking %s queries/s"
     % (f(g(sec, "split2", "scores_ms")), f(g(sec, "split2", "roofline", "achieved"), "%.0f"), f(g(sec, "split2", "roofline", "frac")),
        f(g(sec, "split2", "max_abs_diff_vs_exact_chain"), "%.1e"), f(g(sec, "split2", "top100_slot_agreement_with_exact"), "%.4f"),
        f(g(sec, "split2", "queries_per_s_with_the_fp32_ranking"), "%.0f")), "`secondary_configs.split2`"),
    ("configs[4]: fp16 shard (HBM-bound)", "%s ms = %s GB/s = %s of 8 TB/s; top-100 agreement with fp32 %s"
     % (f(g(sec, "configs4_fp16_shard", "scores_ms")), f(g(sec, "configs4_fp16_shard", "roofline", "achieved"), "%.0f"),
        f(g(sec, "configs4_fp16_shard", "roofline", "frac")), f(g(sec, "configs4_fp16_shard", "top100_slot_agreement_with_fp32"), "%.4f")),
     "`secondary_configs.configs4_fp16_shard`"),
    ("configs[1]: rOxford5k alone (70 x 4 993)", "%s + %s us = %s queries/s" % (f(g(sec, "configs1_roxford5k", "scores_us"), "%.0f"),
                                                                               f(g(sec, "configs1_roxford5k", "rank_us"), "%.0f"),
                                                                               f(g(sec, "configs1_roxford5k", "queries_per_s"), "%.0f")),
     "`secondary_configs.configs1_roxford5k`"),
    ("ONE evaluation's product: the row-major database read in place / index build + multiply / resident index",
     "%s / %s / %s ms, bit-identical %s" % (f(g(sec, "configs2_one_evaluation", "in_place_ms")), f(g(sec, "configs2_one_evaluation", "index_build_plus_multiply_ms")),
                                           f(g(sec, "configs2_one_evaluation", "resident_index_ms")), g(sec, "configs2_one_evaluation", "bit_identical_to_the_index_route")),
     "`secondary_configs.configs2_one_evaluation`"),
    ("ONE whole evaluation on resident descriptors, wall clock (product in place + mAP): default route / literal route",
     "%s / %s ms" % (f(g(sec, "configs2_one_evaluation", "evaluation_ms_default_route"), "%.2f"), f(g(sec, "configs2_one_evaluation", "evaluation_ms_literal_route"), "%.2f")),
     "`secondary_configs.configs2_one_evaluation`, `profiles/r04_eval_path.md`"),
    ("exact top-100 of 1 M x 70 (serving form)", "%s ms" % f(g(sec, "configs2_top100", "topk_ms")), "`secondary_configs.configs2_top100`"),
    ("whitening learning, float64: Gram / projection (D = 2048, n = 20 000)", "%s ms = %s of the f64 MFMA peak / %s ms = %s"
     % (f(g(sec, "whitening_learning_f64", "gram_ms"), "%.2f"), f(g(sec, "whitening_learning_f64", "roofline_gram", "frac")),
        f(g(sec, "whitening_learning_f64", "project_ms"), "%.2f"), f(g(sec, "whitening_learning_f64", "roofline_project", "frac"))),
     "`secondary_configs.whitening_learning_f64`"),
    ("CLAHE input conversion (4 x 1024x768; parity unpinned)", "%s ms per batch = %s of 8 TB/s (plain conversion %s ms)"
     % (f(g(sec, "clahe_preprocess", "ms_per_batch"), "%.4f"), f(g(sec, "clahe_preprocess", "roofline", "frac")),
        f(g(sec, "clahe_preprocess", "plain_u8_to_chw_ms_per_batch"), "%.4f")), "`secondary_configs.clahe_preprocess`"),
    ("descriptors/s: ResNet101-GeM, 3 scales + whitening, a whole warm list of 1 024 JPEG files of 16 sizes",
     "%s descriptors/s (%s ms per image); steady-state estimate %s; loader alone %s images/s"
     % (f(ds.get("value"), "%.1f"), f(ds.get("ms_per_image"), "%.2f"), f(g(ds, "steady_state_estimate", "descriptors_per_s"), "%.1f"),
        f(ds.get("loader_only_images_per_s"), "%.0f")), "`descriptors_per_s`"),
    ("  its roofline: trunk convolutions (MIOpen + own expand 1x1), whole list / steady state",
     "%s TFLOP/s = %s of the fp32 MFMA peak / %s = %s" % (f(g(ds, "roofline", "achieved"), "%.1f"), f(g(ds, "roofline", "frac")),
                                                         f(g(ds, "roofline", "achieved_at_steady_state"), "%.1f"), f(g(ds, "roofline", "frac_at_steady_state"))),
     "`descriptors_per_s.roofline`"),
    ("  CPU reference beside it (the reference-style loop: batch 1, 3 scales, torch CPU ops): 3 threads / 16 threads",
     "%s / %s descriptors/s" % (f(g(ds, "cpu_baseline", "legs", "threads_3", "descriptors_per_s"), "%.2f"),
                               f(g(ds, "cpu_baseline", "legs", "threads_16", "descriptors_per_s"), "%.2f")), "`descriptors_per_s.cpu_baseline`"),
    ("  resident single shape / VGG16", "%s / %s descriptors/s" % (f(ds.get("resident_single_shape_descriptors_per_s"), "%.1f"),
                                                                  f(ds.get("vgg16_resident_single_shape_descriptors_per_s"), "%.1f")), "`descriptors_per_s`"),
]
lines = ["<!-- NUMBERS:BEGIN (generated by tools/design_numbers.py %s from profiles/%s_bench_line.json + profiles/%s_traffic.json; do not edit by hand) -->" % (tag, tag, tag),
         "", "| what | measured on one MI355X (round %s) | field of the bench line |" % tag.lstrip("r0"), "|---|---|---|"]
lines += ["| %s | %s | %s |" % row for row in rows]
lines += ["", "<!-- NUMBERS:END -->"]
path = os.path.join(root, "DESIGN.md")
text = open(path).read()
a, z = text.index("<!-- NUMBERS:BEGIN"), text.index("<!-- NUMBERS:END -->") + len("<!-- NUMBERS:END -->")
open(path, "w").write(text[:a] + "\n".join(lines) + text[z:])
print("\n".join(lines))
