"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE counter CSVs of tools/roofline_pmc.py -> profiles/r01_roofline_pmc*.json.
usage: pmc_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <rows> [split]
HBM bytes per launch = 2 x median FETCH_SIZE KB (gfx950 counts 64-byte units as KB of 32-byte: MI355X_MICROARCH.md) + median
WRITE_SIZE KB (the median: a few launches carry the write-back of the weight-packing kernels that ran just before)."""
import csv, json, statistics, sys

fetch_csv, write_csv, out, rows = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
split = len(sys.argv) > 5 and sys.argv[5] == "split"


def column(path, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if "::skinny_kernel<" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return vals


res = {}
for name, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
    v = column(path, name)
    res[name] = {"launches": len(v), "mean_KB": statistics.mean(v), "median_KB": statistics.median(v), "min_KB": min(v), "max_KB": max(v)}
res["kernel"] = (f"skinny_kernel<bf16,TPW=2,no prologue,U=7,RS> (decode gate/up, split RMSNorm, + SwiGLU), {rows} rows, 24 layers cycled" if split else
                 f"skinny_kernel<bf16,TPW=2,norm,TPR={32 if rows <= 8 else 16},U=7> (decode gate/up + RMSNorm prologue + SwiGLU), {rows} rows, 24 layers cycled")
res["hbm_bytes_per_launch"] = (2 * res["FETCH_SIZE"]["median_KB"] + res["WRITE_SIZE"]["median_KB"]) * 1024
res["note"] = ("separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over tools/roofline_pmc.py; FETCH_SIZE doubled per the guide "
               "(gfx950 counts 64-byte units as KB of 32-byte); medians")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
