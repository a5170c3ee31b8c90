"""Aggregate a rocprofv3 --pmc counter_collection.csv into per-kernel averages (keeps profiles/ small)."""
import csv, json, sys, collections
path, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0, 0.0]))
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or r.get("kernel_name")
        cn, cv = r.get("Counter_Name") or r.get("Counter Name"), r.get("Counter_Value") or r.get("Counter Value")
        if name is None or cn is None:
            continue
        a = acc[name][cn]
        a[0] += float(cv); a[1] += 1; a[2] = max(a[2], float(cv))
# "max": the largest single dispatch -- the whole-sequence launch of a kernel that the step also launches over shorter time windows
res = {k: {c: {"avg": v[0] / v[1], "dispatches": v[1], "max": v[2]} for c, v in d.items()} for k, d in acc.items()}
for d in res.values():      # MI355X: GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"]["avg"] > 0:
        d["mfma_busy_frac"] = {"avg": d["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / (d["GRBM_GUI_ACTIVE"]["avg"] / 8.0 * 1024.0), "dispatches": 0}
top = dict(sorted(res.items(), key=lambda kv: -max(x["avg"] * max(x["dispatches"], 0) for x in kv[1].values()))[:48])
json.dump(top, open(out, "w"), indent=1)
for k, d in list(top.items())[:8]:
    print(k[:90], {c: (round(v["avg"], 2), v["dispatches"]) for c, v in d.items()})
