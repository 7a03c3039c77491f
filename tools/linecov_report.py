"""Merge cov_*.json files written by tools/linecov/sitecustomize.py and list the lines of the host surface that no test entered.

    python tools/linecov_report.py DIR [DIR ...] [--md profiles/r05_host_branches.md]

Executable lines come from the compiled code objects (dis.findlinestarts, nested functions included); `def` / `class` /
decorator lines and docstrings count as executed when their module was imported."""
import dis
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["mdir_amd/%s.py" % m for m in ("score", "datasets", "network", "networks", "wrapper", "validation", "stages", "scenario",
                                        "events", "evaluate", "whiten", "layers", "mining", "cirtorch_format", "sharded", "graphs",
                                        "resample", "jpeg", "ops", "backbones", "trace", "_lib")] + ["eval.py"]


def executable_lines(path):
    with open(path) as f:
        code = compile(f.read(), path, "exec")
    lines, stack = set(), [code]
    while stack:
        c = stack.pop()
        lines.update(l for _, l in dis.findlinestarts(c) if l)
        stack.extend(k for k in c.co_consts if hasattr(k, "co_code"))
    return lines


def ranges(nums):
    out, start, prev = [], None, None
    for n in sorted(nums):
        if start is None:
            start = prev = n
        elif n == prev + 1:
            prev = n
        else:
            out.append((start, prev))
            start = prev = n
    if start is not None:
        out.append((start, prev))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    hit = {}
    for d in args:
        for fn in glob.glob(os.path.join(d, "cov_*.json")):
            for k, v in json.load(open(fn)).items():
                hit.setdefault(k, set()).update(v)
    report = {}
    for rel in FILES:
        path = os.path.join(ROOT, rel)
        ex = executable_lines(path)
        miss = ex - hit.get(rel, set())
        src = open(path).read().splitlines()
        report[rel] = {"executable": len(ex), "missed": len(miss),
                       "ranges": [(a, b, src[a - 1].strip()[:110]) for a, b in ranges(miss)]}
    for rel, r in report.items():
        print("%-28s %4d executable, %3d not entered" % (rel, r["executable"], r["missed"]))
        for a, b, text in r["ranges"]:
            print("    %s  %s" % (("%d" % a) if a == b else "%d-%d" % (a, b), text))
    json.dump(report, open(os.path.join(args[0], "report.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
