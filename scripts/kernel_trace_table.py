"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): calls, average and total duration -- which INSTANCES of a
templated kernel (e.g. which GEMM shapes) carry the time.  usage: python scripts/kernel_trace_table.py <kernel_trace.csv> [min_us]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
agg = collections.OrderedDict()
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    short = name.replace("void ", "").replace("(anonymous namespace)::", "")
    short = short.split("(")[0][:70]
    key = (short, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
print("total kernel time %.1f ms over %d dispatches" % (tot / 1e3, len(rows)))
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if t < min_us:
        continue
    print("%9.1f us total %5d calls %9.1f us avg  grid %s x %s x %s wg %s  %s" % (t, n, t / n, key[1], key[2], key[3], key[4], key[0]))
