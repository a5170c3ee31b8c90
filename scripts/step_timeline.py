"""Ordered timeline of ONE training step from a rocprofv3 --kernel-trace CSV: every dispatch between two consecutive radam_k
launches with its start offset, duration and the idle gap in front of it -- shows where the GPU waits for the host (launch-bound
stretches) and which kernels sit between the recurrences.  usage: python scripts/step_timeline.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short,
               "%sx%sx%s" % (r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))))
ev.sort()
marks = [i for i, e in enumerate(ev) if e[2].startswith(("radam_k", "radam_dev_k"))]
if len(marks) < back + 1:
    print("not enough optimizer steps in the trace (%d radam_k launches)" % len(marks))
    sys.exit(0)
lo, hi = marks[-back - 1] + 1, marks[-back] + 1
step = ev[lo:hi]
t0 = ev[lo - 1][1]                      # end of the previous step's optimizer kernel
busy = sum(e[1] - e[0] for e in step)
span = step[-1][1] - t0
print("step of %d dispatches: span %.2f ms, kernel-busy %.2f ms, idle %.2f ms" % (len(step), span / 1e6, busy / 1e6, (span - busy) / 1e6))
prev = t0
gaps = collections.Counter()
agg = collections.OrderedDict()
for s, e, n, g in step:
    gap = (s - prev) / 1e3
    b = "<2us" if gap < 2 else "2-5us" if gap < 5 else "5-10us" if gap < 10 else "10-30us" if gap < 30 else ">30us"
    gaps[b] += max(gap, 0.0)
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
    prev = max(prev, e)
print("idle by gap size (us):", {k: round(v, 1) for k, v in gaps.items()})
print("--- by kernel (this step)")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%9.1f us %5d x  %s" % (t, c, n))
print("--- gaps >= 20 us (offset us, gap us, kernel before -> kernel after)")
prev, pn = t0, "radam_k (previous step)"
for s_, e_, n_, g_ in step:
    if (s_ - prev) / 1e3 >= 20.0:
        print("%9.1f %8.1f  %s -> %s" % ((s_ - t0) / 1e3, (s_ - prev) / 1e3, pn, n_))
    if e_ > prev:
        prev, pn = e_, n_
print("--- timeline (offset us, gap us, dur us, kernel, grid)")
prev = t0
for s, e, n, g in step:
    print("%9.1f %7.1f %8.1f  %s  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n, g))
    prev = max(prev, e)
