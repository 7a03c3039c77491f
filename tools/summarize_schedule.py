"""gpurun_out/schedule_r05 (tools/pmc_schedule.sh) -> profiles/r05_scores_schedule.md: the exact similarity kernel's two
consumer schedules, SQ counters per launch and the in-process A/B timings."""
import collections
import csv
import os
import re
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "schedule_r05")
rows = {}
for pipe in (0, 1):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(SRC, "pmc_pipe%d.csv" % pipe))):
        if r["Counter_Name"] == "Counter_Name":
            continue
        key = r["Dispatch_Id"] + "@" + r["Start_Timestamp"]
        per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[key]["_ns"] = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"])]
        per[key]["_name"] = [r["Kernel_Name"].split("(")[0]]
    launches = [v for v in per.values() if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_VALU_MFMA_BUSY_CYCLES")]
    cyc = [v["GRBM_GUI_ACTIVE"][0] / 8.0 for v in launches]
    rows[pipe] = {
        "kernel": launches[0]["_name"][0].replace("void ", ""), "launches": len(launches),
        "ms": statistics.mean(v["_ns"][0] for v in launches) / 1e6,
        "ghz": statistics.mean(c / v["_ns"][0] for c, v in zip(cyc, launches)),
        "busy": statistics.mean(v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 1024 / c for c, v in zip(cyc, launches)),
        "wait": statistics.mean(v["SQ_WAIT_ANY"][0] / v["SQ_WAVE_CYCLES"][0] for v in launches),
        "wait_inst": statistics.mean(v["SQ_WAIT_INST_ANY"][0] / v["SQ_WAVE_CYCLES"][0] for v in launches),
        "conflict": statistics.mean(v["SQ_LDS_BANK_CONFLICT"][0] for v in launches),
        "cycles": statistics.mean(cyc)}
ab = collections.defaultdict(list)
for line in open(os.path.join(SRC, "ab.log")):
    m = re.match(r"(gaussian unit rows|all zero)\s+round \d+: shipped ([\d.]+) ms\s+pipelined ([\d.]+) ms.*bit-equal (\w+)", line)
    if m:
        ab[m.group(1)].append((float(m.group(2)), float(m.group(3)), m.group(4) == "True"))
out = os.path.join(ROOT, "profiles", "r05_scores_schedule.md")
with open(out, "w") as f:
    f.write("# r05: the f64 GEMM's schedule in the exact fp32 similarity kernel (VERDICT round 4, item 3)\n\n"
            "`scores_lc_kernel<QT=4,R=2,KC=2,NSTAGE=3,QR=1>` at N = 1 004 993, Q = 70, D = 2048 in two consumer schedules, same\n"
            "library, same box (`tools/pmc_schedule.sh`, `tools/scores_pipe_probe.py`; `MDX_SCORES_PIPE` is read per launch):\n\n"
            "* **r04 schedule**: one raw `s_barrier` at the top of every 32-k chunk, the chunk's LDS reads behind it, all 64 + 16 MFMAs\n"
            "  behind those; a second `lgkmcnt(0)` between the chunk's two k-blocks.\n"
            "* **pipelined (PIPE)**: what took `gemm_f64_lc_kernel` from 69 %% to 83 %% pipe occupancy -- the next k-block's first operands\n"
            "  (database tiles + query tile 0) are read a block ahead, query tiles 1-3 during the previous block's last step (each into\n"
            "  the register its last MFMA has just released: a full second operand set does not fit in 128 registers), and the stage\n"
            "  hand-over (`lgkmcnt(0)`, `s_barrier`, first reads of the new stage) sits in front of a stage's LAST step, so 8 + 8 MFMAs\n"
            "  run through the barrier.  124 VGPRs, no scratch, two workgroups per CU as before; k-ascending chain per output kept.\n\n"
            "## SQ counter pass (rocprofv3 --pmc, %d + %d launches, gaussian unit rows)\n\n"
            "| schedule | launch ms (counter pass) | cycles per launch (GRBM_GUI_ACTIVE/8) | sustained clock GHz | MFMA pipe busy | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | LDS bank conflicts |\n|---|---|---|---|---|---|---|---|\n"
            % (rows[0]["launches"], rows[1]["launches"]))
    for pipe, name in ((0, "r04"), (1, "pipelined")):
        r = rows[pipe]
        f.write("| %s | %.3f | %.4g | %.3f | %.1f %% | %.3f | %.3f | %.0f |\n"
                % (name, r["ms"], r["cycles"], r["ghz"], 100 * r["busy"], r["wait"], r["wait_inst"], r["conflict"]))
    f.write("\nKernel names: `%s` / `%s`.\n\n## A/B timing, HIP events, 20 launches per figure, interleaved in one process (three processes x three rounds)\n\n"
            "| operands | r04 ms (min / median / max) | pipelined ms (min / median / max) | median change | bit-equal |\n|---|---|---|---|---|\n"
            % (rows[0]["kernel"], rows[1]["kernel"]))
    for name, v in ab.items():
        a, b = [x[0] for x in v], [x[1] for x in v]
        f.write("| %s | %.3f / %.3f / %.3f | %.3f / %.3f / %.3f | %+.1f %% | %s (%d of %d) |\n"
                % (name, min(a), statistics.median(a), max(a), min(b), statistics.median(b), max(b),
                   100 * (statistics.median(b) - statistics.median(a)) / statistics.median(a), all(x[2] for x in v), sum(x[2] for x in v), len(v)))
print(open(out).read())
with open(out, "a") as f:
    f.write("\n## Reading\n\n"
            "The move that bought the f64 GEMM 14 points of pipe occupancy buys this kernel **%.1f points** (%.1f %% -> %.1f %%) and\n"
            "%.1f %% of its cycles; on real rows the clock gives %.1f %% of that back (%.3f -> %.3f GHz: the power envelope, as DESIGN\n"
            "section 8 argued), so the launch gets %.1f %% shorter on real rows and %.1f %% on all-zero operands, bit for bit the same scores.\n"
            "Why so little here: the f64 kernel ran ONE consumer wave per SIMD, so every barrier + LDS round trip was an empty pipe; this\n"
            "kernel has TWO consumer waves per SIMD (two 78-KiB workgroups per CU) that already fill each other's bubbles -- the\n"
            "r02 stamps showed the waves waiting at barriers for 27 %% of the loop while the pipe stayed 80 %% busy.  What is left of\n"
            "the pipe's idle fifth is not the consumer's read-after-barrier latency (removed here) but both waves of a SIMD waiting for\n"
            "the SAME event: a stage of the shared stream that has not landed, and the 13-us epilogue + ring fill at the ends of a\n"
            "workgroup's 180-us life (7 %% of it).  The pipelined form is the default from round 5 on (`MDX_SCORES_PIPE=0` restores\n"
            "the round-4 schedule per launch); the row-major in-place variant (`RM`) keeps the round-4 schedule.\n"
            % (100 * (rows[1]["busy"] - rows[0]["busy"]), 100 * rows[0]["busy"], 100 * rows[1]["busy"],
               100 * (1 - rows[1]["cycles"] / rows[0]["cycles"]), 100 * (1 - rows[1]["ghz"] / rows[0]["ghz"]), rows[0]["ghz"], rows[1]["ghz"],
               -100 * (statistics.median([x[1] for x in ab["gaussian unit rows"]]) / statistics.median([x[0] for x in ab["gaussian unit rows"]]) - 1),
               -100 * (statistics.median([x[1] for x in ab["all zero"]]) / statistics.median([x[0] for x in ab["all zero"]]) - 1)))
print(open(out).read()[-1600:])
