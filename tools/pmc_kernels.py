"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: calls, total duration, and every collected counter summed.
usage: pmc_kernels.py <counter_collection.csv> <out.csv>"""
import collections, csv, re, sys

src, out = sys.argv[1], sys.argv[2]
disp = {}
for r in csv.DictReader(open(src)):
    k = (r["Dispatch_Id"])
    d = disp.setdefault(k, {"name": r["Kernel_Name"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
    d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.OrderedDict()
names = sorted({c for d in disp.values() for c in d["c"]})
for d in disp.values():
    n = re.sub(r"\(anonymous namespace\)::", "", d["name"])
    a = agg.setdefault(n, {"calls": 0, "dur": 0, **{c: 0.0 for c in names}})
    a["calls"] += 1
    a["dur"] += d["dur"]
    for c, v in d["c"].items():
        a[c] += v
rows = sorted(agg.items(), key=lambda kv: -kv[1]["dur"])
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDurationNs"] + names)
    for n, a in rows:
        w.writerow([n, a["calls"], a["dur"]] + [round(a[c], 1) for c in names])
for n, a in rows[:8]:
    print(f'{a["dur"] / 1e6:9.2f} ms {a["calls"]:6d} x  {n[:70]}  ' + "  ".join(f"{c}={a[c]:.3g}" for c in names))
