"""Print the last N mdx kernels of a rocprofv3 kernel trace csv: duration and gap to the previous kernel."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mdx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -30:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-58s dur %7.1f us  gap %6.1f us  grid %s" % (r["Kernel_Name"][:58], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0,
                                                        r.get("Grid_Size_X", r.get("Grid_Size"))))
    prev = e
