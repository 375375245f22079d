# round 6: what the chip does while several steps are in flight - from a rocprofv3 --kernel-trace run of bench.py:
#   * the fraction of the traced span during which at least one kernel runs (union of the kernels' intervals), and the time-weighted number of kernels in flight
#   * per kernel family: summed duration, and the part of it during which it ran ALONE / beside 1 / 2 / 3+ other kernels
#   * (with --hip-trace in the same run) the host API calls by total time
#   python tools/r06/busy.py <rocprof output dir> [first_ms last_ms]
import csv, glob, os, sys
from collections import defaultdict


def find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))


def short(n):
    n = n.replace("void ", "").replace("necat::", "")
    return n.split("(")[0][:44]


d = sys.argv[1]
rows = []
for f in find(d, "kernel_trace.csv"):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
t0 = iv[0][0]
if len(sys.argv) > 3:
    a, b = t0 + int(float(sys.argv[2]) * 1e6), t0 + int(float(sys.argv[3]) * 1e6)
    iv = [(max(s, a), min(e, b), n) for s, e, n in iv if e > a and s < b]
span0, span1 = min(s for s, _, _ in iv), max(e for _, e, _ in iv)
ev = []
for i, (s, e, n) in enumerate(iv):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set()
last = ev[0][0]
by_depth = defaultdict(int)
alone = defaultdict(lambda: defaultdict(int))
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        k = len(active)
        by_depth[k] += dt
        for j in active:
            alone[iv[j][2]][min(k - 1, 3)] += dt
    last = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
span = span1 - span0
print("traced span %.2f ms, %d kernels" % (span / 1e6, len(iv)))
print("no kernel running: %.1f %% | kernels in flight (time-weighted): %s" % (
    100.0 * by_depth[0] / span, ", ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(by_depth.items()) if k and v * 200 > span)))
# the union's busy fraction per 50 ms of the span (the bench's timed region is where it is highest)
bins = defaultdict(int)
BIN = 50_000_000
cur_s, cur_e = None, None
merged = []
for s, e, n in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            merged.append((cur_s, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
merged.append((cur_s, cur_e))
for s, e in merged:
    while s < e:
        b = (s - span0) // BIN
        x = min(e, span0 + (b + 1) * BIN)
        bins[b] += x - s
        s = x
print("busy %% per 50 ms bin: " + " ".join("%d" % round(100.0 * bins[b] / BIN) for b in range((span + BIN - 1) // BIN)))
tot = defaultdict(int)
for s, e, n in iv:
    tot[n] += e - s
print("%-46s %9s  %6s %6s %6s %6s" % ("kernel", "sum ms", "alone", "+1", "+2", "+3.."))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:18]:
    a = alone[n]
    print("%-46s %9.2f  %5.0f%% %5.0f%% %5.0f%% %5.0f%%" % (n, v / 1e6, *(100.0 * a[k] / max(1, v) for k in range(4))))
api = defaultdict(lambda: [0, 0])
for f in find(d, "hip_api_trace.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            x = api[r["Function"]]
            x[0] += 1; x[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
if api:
    print("host API calls by total time:")
    for fn, (c, ns) in sorted(api.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-36s %8d calls %10.2f ms" % (fn, c, ns / 1e6))
