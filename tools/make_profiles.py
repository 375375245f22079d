#!/usr/bin/env python3
"""Summarise rocprofv3 outputs (written under gpurun_out/ on the GPU box) into the tracked files under
profiles/.

  python tools/make_profiles.py stats  <dir> <out.md>  "<command>"   # --kernel-trace --stats run
  python tools/make_profiles.py pmc    <dir_fetch> <dir_write> <out.json>   # two --pmc passes

The PMC values are per-launch averages of FETCH_SIZE / WRITE_SIZE (KB as the counters report them; the
gfx950 correction of MI355X_MICROARCH.md's HBM section is applied by the reader, bench.py).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits


def stats(d, out, cmd):
    rows = []
    for f in find(d, "kernel_stats.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        a = agg[r["Name"]]
        a[0] += int(r["Calls"]); a[1] += int(float(r["TotalDurationNs"]))
    total = sum(a[1] for a in agg.values()) or 1
    with open(out, "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats\n\nCommand: `%s`\n\n"
                 "Durations are per kernel, start to end, and the extension runs three chains side by side (list A on one stream, list B on\n"
                 "two more): the totals add up to more than the wall time, and a small kernel of the list-B chain (`k_items_*`, `k_ext_frag<13, 25>`,\n"
                 "the 794-column DP / walk kernels) that starts while list A's issue-bound DP kernel fills the chip is timed with its wait for\n"
                 "issue slots included (`profiles/r02_round_timeline.txt` shows one step launch by launch).\n\n"
                 "| kernel | calls | total ns | avg ns | %% |\n|---|---|---|---|---|\n" % cmd)
        for name, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fh.write("| `%s` | %d | %d | %d | %.2f |\n" % (name, calls, ns, ns // max(1, calls), 100.0 * ns / total))


def pmc(dirs, out):
    res = defaultdict(dict)
    for d in dirs:
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for f in find(d, "counter_collection.csv"):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = r["Kernel_Name"]
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    disp[k].add(r["Dispatch_Id"])
        for k, cs in acc.items():
            n = max(1, len(disp[k]))
            res[k]["launches"] = n
            for c, v in cs.items():
                res[k][c + "_KB_per_launch"] = v / n
    with open(out, "w") as fh:
        json.dump(dict(sorted(res.items())), fh, indent=1)


def counters(dirs, out):
    """raw per-kernel sums of every counter of several --pmc passes (+ dispatch counts), tolerant of failed passes"""
    res = defaultdict(dict)
    for d in dirs:
        try:
            files = find(d, "counter_collection.csv")
        except SystemExit:
            continue
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for f in files:
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = r["Kernel_Name"]
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    disp[k].add(r["Dispatch_Id"])
        for k, cs in acc.items():
            res[k].setdefault("launches", len(disp[k]))
            for c, v in cs.items():
                res[k][c] = v
    with open(out, "w") as fh:
        json.dump(dict(sorted(res.items())), fh, indent=1)


def timeline(d, out, header):
    """the kernels of the LAST bench step of a --kernel-trace run, launch by launch: start (ms from the step's first kernel),
    duration, grid, queue, name - the step starts at the last k_part_hist of the trace"""
    rows = []
    for f in find(d, "kernel_trace.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("necat::k_part_hist") or "k_part_hist" in r["Kernel_Name"]]
    if not starts:
        raise SystemExit("no k_part_hist in the trace")
    # the last step that ran the extension (the -j 0 extras of bench.py come after the timed steps)
    bounds = starts + [len(rows)]
    segs = [(bounds[i], bounds[i + 1]) for i in range(len(starts)) if any("k_ext_init" in r["Kernel_Name"] for r in rows[bounds[i]:bounds[i + 1]])]
    a, b = segs[-1] if segs else (starts[-1], len(rows))
    rows = rows[a:b]
    t0 = int(rows[0]["Start_Timestamp"])
    queues = {}
    with open(out, "w") as fh:
        fh.write("# %s\n# start ms | duration us | grid threads | queue | kernel\n" % header)
        for r in rows:
            q = queues.setdefault(r["Queue_Id"], len(queues) + 1)
            name = r["Kernel_Name"].replace("necat::", "").replace("void ", "")
            name = name.split("(")[0]
            grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            fh.write("%8.3f %9.1f %10d q%d %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, grid, q, name))


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "counters":
        counters(sys.argv[2:-1], sys.argv[-1])
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2:-1], sys.argv[-1])
