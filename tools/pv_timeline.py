#!/usr/bin/env python3
"""tools/pv_timeline.py <kernel_trace.csv> [chunk] — the last phase-vocoder call of a rocprofv3 --kernel-trace run as a timeline:
span, union of kernel intervals (machine not idle), idle gaps, time per kernel name, how much of the span two kernels overlap,
and the dispatches of one chunk (queue, start offset, duration).  Used to see what the chunked pipeline (capi_pv.cpp) leaves bare."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.replace("mx::(anonymous namespace)::", "").replace("void ", "").split("(")[0],
                 r.get("Queue_Id", "?")))
rows.sort()
# calls: clusters of pv_ kernels separated by more than 2 ms
pv = [x for x in rows if x[2].startswith("pv_")]
calls, cur = [], [pv[0]]
for x in pv[1:]:
    if x[0] - max(y[1] for y in cur[-50:]) > 2_000_000:
        calls.append(cur)
        cur = []
    cur.append(x)
calls.append(cur)
c = calls[-1]
t0, t1 = c[0][0], max(x[1] for x in c)
print(f"calls {len(calls)}; last call: {len(c)} dispatches, span {(t1 - t0) / 1e6:.3f} ms")
ev = sorted([(x[0], 1) for x in c] + [(x[1], -1) for x in c])
depth, last, busy, over = 0, t0, 0, 0
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += d
    last = t
print(f"union of kernel intervals {busy / 1e6:.3f} ms, idle {((t1 - t0) - busy) / 1e6:.3f} ms, two or more kernels in flight {over / 1e6:.3f} ms")
per = collections.defaultdict(lambda: [0, 0])
for x in c:
    per[x[2]][0] += 1
    per[x[2]][1] += x[1] - x[0]
for k, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:32s} x{n:4d}  total {d / 1e6:8.3f} ms  avg {d / n / 1e3:8.1f} us")
# one chunk: from the (k)th pv_analysis to the next
an = [i for i, x in enumerate(c) if x[2] == "pv_analysis"]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(an) // 2
if len(an) > k + 2:
    lo, hi = an[k], an[k + 2]
    base = c[lo][0]
    print(f"dispatches from analysis {k} to analysis {k + 2} (offsets in us from the first):")
    for x in c[lo:hi + 1]:
        print(f"  q{x[3]:>3s} {x[2]:32s} start {(x[0] - base) / 1e3:9.1f}  dur {(x[1] - x[0]) / 1e3:8.1f}")
