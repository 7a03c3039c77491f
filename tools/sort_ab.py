"""A/B of library builds on one device: runs tools/quick_bench.py under each MDIR_AMD_LIB, interleaved rounds."""
import os, subprocess, sys
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "quick_bench.py")],
                             env=dict(os.environ, MDIR_AMD_LIB=os.path.abspath(l)), capture_output=True, text=True).stdout
        t = {ln.split()[0]: float(ln.split()[1]) for ln in out.splitlines() if ln.startswith(("scores", "rank_full"))}
        res[l].append(t)
for l in libs:
    print(l, " scores ms:", [r.get("scores") for r in res[l]], " rank ms:", [r.get("rank_full") for r in res[l]])
