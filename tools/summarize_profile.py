#!/usr/bin/env python3
"""Digest gpurun_out/profile_<tag>/ (tools/profile_round.sh) into tracked files:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (mdx kernels + top others)
   profiles/<tag>_traffic.json       HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes
   profiles/<tag>_summary.md         human-readable table
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of a wide
coalesced streaming read -> doubled here; WRITE_SIZE is exact for 16-B streaming stores.  Both in KiB."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "profile_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
keep = [r for r in rows if "mdx::" in r["Name"]]
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in keep + [r for r in rows if "mdx::" not in r["Name"]][:8]:
        w.writerow(r)


def short(name):
    n = name.replace("void ", "")
    return n[:n.index("(")] if "(" in n else n


def pmc(fn):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    p = os.path.join(src, fn)
    if not os.path.exists(p):
        return d
    for r in csv.DictReader(open(p)):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


fetch, write, sq = pmc("pmc_FETCH_SIZE.csv"), pmc("pmc_WRITE_SIZE.csv"), pmc("pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    fv = [v for v in fetch[k].get("FETCH_SIZE", []) if v > 0]
    wv = [v for v in write[k].get("WRITE_SIZE", []) if v > 0]
    rd = 2.0 * 1024 * (sum(fv) / len(fv)) if fv else 0.0
    wr = 1024 * (sum(wv) / len(wv)) if wv else 0.0
    traffic[k] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
sk = [k for k in traffic if "::scores_" in k and "gather" not in k]
clock = mfma_busy = None
for k in sq:            # sustained clock and MFMA pipe occupancy of the similarity kernel in the SQ counter pass
    if "::scores_" in k and "gather" not in k and sq[k].get("GRBM_GUI_ACTIVE") and sq[k].get("SQ_VALU_MFMA_BUSY_CYCLES"):
        cyc = sum(sq[k]["GRBM_GUI_ACTIVE"]) / len(sq[k]["GRBM_GUI_ACTIVE"]) / 8.0
        durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                for r in csv.DictReader(open(os.path.join(src, "pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv")))
                if short(r["Kernel_Name"]) == k and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
        if durs and cyc > 1e6:
            clock = cyc / (sum(durs) / len(durs))                   # cycles per ns = GHz
            mfma_busy = sum(sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024 / cyc
sys.path.insert(0, root)
from bench import rank_source_sha16, ranking_traffic, scores_source_sha16      # tie the figures below to the sources they were measured on
out = {"note": "HBM bytes per launch; FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE x1; KiB -> bytes",
       "scores_kernel_source_sha16": scores_source_sha16(),
       "rank_source_sha16": rank_source_sha16(),
       "ranking_hbm_bytes_per_ranking": ranking_traffic({"per_kernel": traffic}),
       "per_kernel": traffic,
       "scores_kernel_hbm_bytes_per_launch": traffic[sk[0]]["total_bytes"] if sk else None,
       "scores_kernel_sustained_clock_ghz": round(clock, 3) if clock else None,
       "scores_kernel_mfma_pipe_busy": round(mfma_busy, 4) if mfma_busy else None}
# the kernel-trace averages of the same command (bench.py --profile): what `roofline.frac_rocprof` of the bench line is computed from
_sk_rows = sorted((r for r in keep if "::scores_lc_kernel" in r["Name"]), key=lambda r: -float(r["TotalDurationNs"]))
if _sk_rows:
    out["scores_kernel_trace_avg_us"] = round(float(_sk_rows[0]["AverageNs"]) / 1e3, 2)
    out["scores_kernel_trace_name"] = short(_sk_rows[0]["Name"])
    out["scores_kernel_trace_calls"] = int(_sk_rows[0]["Calls"])
out["ranking_trace_sum_us"] = round(sum(float(r["AverageNs"]) / 1e3 * (4 if "sort_scan_kernel" in r["Name"] else 1)
                                        for r in keep if "::sort_" in r["Name"]), 2)
# the split-precision kernel's own passes (tools/profile_round.sh: tools/split_bench.py under --pmc)
f3, w3, s3 = pmc("pmc_split3_FETCH_SIZE.csv"), pmc("pmc_split3_WRITE_SIZE.csv"), pmc("pmc_split3_SQ_VALU_MFMA_BUSY_CYCLES.csv")
for k in f3:
    if "scores_split3" in k or "scores_split2" in k:
        key = "split3_kernel" if "scores_split3" in k else "split2_kernel"
        rd = 2.0 * 1024 * sum(f3[k]["FETCH_SIZE"]) / len(f3[k]["FETCH_SIZE"])
        wr = 1024 * sum(w3[k]["WRITE_SIZE"]) / len(w3[k]["WRITE_SIZE"]) if w3.get(k, {}).get("WRITE_SIZE") else 0.0
        out[key] = {"name": k, "read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr}
        if s3.get(k, {}).get("GRBM_GUI_ACTIVE"):
            cyc = sum(s3[k]["GRBM_GUI_ACTIVE"]) / len(s3[k]["GRBM_GUI_ACTIVE"]) / 8.0
            durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    for r in csv.DictReader(open(os.path.join(src, "pmc_split3_SQ_VALU_MFMA_BUSY_CYCLES.csv")))
                    if short(r["Kernel_Name"]) == k and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
            if durs:
                out[key].update(sustained_clock_ghz=round(cyc / (sum(durs) / len(durs)), 3),
                                            ms_in_the_counter_pass=round(sum(durs) / len(durs) / 1e6, 4),
                                            mfma_pipe_busy=round(sum(s3[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(s3[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024 / cyc, 4),
                                            counters={n: sum(v) / len(v) for n, v in s3[k].items()})
json.dump(out, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)

bench = {}
for line in open(os.path.join(src, "stats_bench.log")):
    if line.startswith("{"):
        bench = json.loads(line)
with open(os.path.join(dst, tag + "_summary.md"), "w") as f:
    f.write("# rocprofv3 summary, round %s\n\n" % tag)
    f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --profile` (the headline loop alone) "
            "(N=1 004 993, Q=70, D=2048 fp32, 20 timed + 3 warm-up steps); PMC passes in separate runs "
            "(`tools/profile_round.sh`).\n\n")
    if bench:
        f.write("bench line under the profiler: %.1f queries/s, %.4f ms/step; scores kernel by HIP events %.4f ms "
                "(%.1f TFLOP/s = %.1f %% of 157.3).\n\n" % (bench["value"], bench["ms_per_step"],
                                                           bench["roofline"]["kernel_ms"], bench["roofline"]["achieved"],
                                                           100 * bench["roofline"]["frac"]))
        # the sort kernels of ONE ranking (4 histograms, 4 scans, 4 scatters) against the bench's own HIP-event figure
        per_rank = 0.0
        for r in keep:
            if "::sort_" in r["Name"]:
                per_rank += float(r["AverageNs"]) / 1e3 * (4 if "sort_scan_kernel" in r["Name"] else 1)
        sk_avg = [float(r["AverageNs"]) / 1e3 for r in keep if "::scores_lc_kernel" in r["Name"] and int(r["Calls"]) >= bench["steps"]]
        if per_rank and bench.get("rank_ms_per_step"):
            f.write("Ranking: the sort kernels' trace averages add up to %.1f us per ranking; bench `rank_ms_per_step` (HIP events around "
                    "the 12 launches) = %.1f us (ratio %.3f).  Similarity kernel: trace average %.1f us, HIP events %.1f us.  "
                    "Step: %.1f us by the wall clock.\n\n" % (per_rank, 1e3 * bench["rank_ms_per_step"], per_rank / (1e3 * bench["rank_ms_per_step"]),
                                                               sk_avg[0] if sk_avg else float("nan"), 1e3 * bench["roofline"]["kernel_ms"], 1e3 * bench["ms_per_step"]))
    f.write("| kernel | calls | avg us | total ms | HBM read MB/launch | HBM write MB/launch |\n|---|---:|---:|---:|---:|---:|\n")
    for r in keep:
        k = short(r["Name"])
        t = traffic.get(k, {})
        f.write("| `%s` | %s | %.1f | %.2f | %s | %s |\n" % (
            k, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6,
            "%.1f" % (t["read_bytes"] / 1e6) if t else "-", "%.1f" % (t["write_bytes"] / 1e6) if t else "-"))
    if sq:
        f.write("\n## SQ counters, scores kernel (per launch, averaged)\n\n")
        for k in sq:
            if "::scores_" in k and "gather" not in k:
                c = {n: sum(v) / len(v) for n, v in sq[k].items()}
                f.write("```\n" + "\n".join("%-28s %.4g" % kv for kv in sorted(c.items())) + "\n```\n")
                if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
                    f.write("\nMFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8) = %.1f %%; "
                            "GRBM_GUI_ACTIVE/8 = %.3g cycles per launch" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, cyc))
                    # sustained clock: cycles of a launch / its duration in the same (counter) pass
                    durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                            for r in csv.DictReader(open(os.path.join(src, "pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv")))
                            if short(r["Kernel_Name"]) == k and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
                    if durs:
                        ms = sum(durs) / len(durs) / 1e6
                        f.write("; launch duration in the counter pass %.3f ms -> sustained clock %.2f GHz (2.4 GHz is what the "
                                "157.3 TFLOP/s peak assumes)" % (ms, cyc / ms / 1e6))
                    busy = pmc("pmc_SQ_BUSY_CYCLES.csv").get(k, {})
                    if busy.get("SQ_BUSY_CYCLES") and busy.get("GRBM_GUI_ACTIVE"):
                        b = sum(busy["SQ_BUSY_CYCLES"]) / len(busy["SQ_BUSY_CYCLES"])
                        g = sum(busy["GRBM_GUI_ACTIVE"]) / len(busy["GRBM_GUI_ACTIVE"])
                        f.write("; SQ_BUSY_CYCLES %.4g per launch = %.2f x GRBM_GUI_ACTIVE of its own pass" % (b, b / g))
                    f.write("\n")
print(open(os.path.join(dst, tag + "_summary.md")).read())
