"""One bench step's extension, round by round, from a rocprofv3 --kernel-trace timeline written by tools/make_profiles.py timeline (profiles/r0N_round_timeline.txt):
list A's chain on its queue - the start of one round's first kernel to the start of the next one's - with the pass, the walk and what is left ("other": fragments or round
bookkeeping, the finishing kernel, the launch gaps), and the excess of every round over the rate of the big rounds (rounds 1 - 8).  profiles/NOTES_r06.md 1.

    python tools/r06/round_table.py profiles/r06_round_timeline.txt > profiles/r06_round_table.txt"""
import sys

rows = []
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        continue
    p = ln.split(None, 4)
    if len(p) < 5:
        continue
    rows.append((float(p[0]), float(p[1]), int(p[2]), p[3], p[4].strip()))
# list A's queue: the one k_myers_ck<8 runs on
qa = next(q for _, _, _, q, k in rows if k.startswith("k_myers_ck<8"))
rounds, cur = [], None
for s, d, g, q, k in rows:
    if q != qa:
        continue
    first = k.startswith("k_ext_frag<8") or k.startswith("k_round_ctl")
    if first:
        if cur:
            rounds.append(cur)
        cur = dict(start=s, n=0, ck=0.0, walk=0.0, fused=0.0, small=d)
    elif cur is None:
        continue
    elif k.startswith("k_myers_ck<"):
        cur["ck"] += d; cur["n"] = g // 8
    elif k.startswith("k_rcwalk"):
        cur["walk"] += d
    elif k.startswith("k_tail_fused<8"):
        cur["fused"] += d; cur["n"] = g // 256; cur["end"] = s + d / 1000
    elif k.startswith("k_traceback<8"):
        cur["small"] += d; cur["end"] = s + d / 1000
if cur:
    rounds.append(cur)
T = []
for i, r in enumerate(rounds):
    nxt = rounds[i + 1]["start"] if i + 1 < len(rounds) else r.get("end", r["start"])
    T.append((nxt - r["start"]) * 1000)
big = [i for i in range(1, min(9, len(rounds))) if rounds[i]["n"] > 150000]
rate = sum(T[i] for i in big) / max(1, sum(rounds[i]["n"] for i in big)) * 1000 if big else 0.0
print("# %s: list A's chain round by round; rate of the big rounds %s: %.2f us per 1000 blocks" % (sys.argv[1], big, rate))
print("# round | blocks | round us | pass us | walk us | fused us | other us (small kernels + gaps) | excess over the big rounds' rate us")
tot = exc = blocks = 0.0
for i, r in enumerate(rounds):
    dp = r["ck"] + r["walk"] + r["fused"]
    e = T[i] - rate * r["n"] / 1000
    print("%5d %8d %9.0f %8.0f %8.0f %8.0f %8.0f %9.0f" % (i, r["n"], T[i], r["ck"], r["walk"], r["fused"], T[i] - dp, e))
    tot += T[i]; exc += e; blocks += r["n"]
print("# %d rounds, %d blocks, %.2f ms; at the big rounds' rate %.2f ms; excess %.2f ms" % (len(rounds), blocks, tot / 1000, rate * blocks / 1e6, exc / 1000))
