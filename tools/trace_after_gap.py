"""Per-kernel totals of a rocprofv3 kernel trace, restricted to what follows the last >0.5 s idle gap."""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
reps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cut = 0
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 5e8:
        cut = i
rows = rows[cut:]
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[r["Kernel_Name"][:100]] += d; cnt[r["Kernel_Name"][:100]] += 1
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
s = sum(tot.values())
print("kernels %d, busy %.2f ms, span %.2f ms (per rep: %.2f / %.2f)" % (len(rows), s / 1e6, span / 1e6, s / 1e6 / reps, span / 1e6 / reps))
for k, v in tot.most_common(18):
    print("%6.2f%% %8.3f ms/rep  x%-5d %s" % (100 * v / s, v / 1e6 / reps, cnt[k] / reps, k))
