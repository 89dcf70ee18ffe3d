"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the SAME command, MI355X_MICROARCH.md §HBM) -> one JSON with the
HBM bytes per launch of every kernel: 2 x FETCH_SIZE KB (gfx950 counts 128-byte requests at 64 B) + WRITE_SIZE KB, medians over launches.
The JSON records the SHA-256 of the kernel source files it was collected on; bench.py only reports a traffic figure whose hash still
matches the source it is running (otherwise `traffic` is null and `traffic_source.status` says stale).

usage: pmc_hbm.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <command string> [source.hip ...]
PMC_UNITS=N in the environment records that the command processed N units (e.g. N vocoder decodes): total_hbm_bytes / N = bytes per unit."""
import collections, csv, hashlib, json, os, re, statistics, subprocess, sys

fetch_csv, write_csv, out, command = sys.argv[1:5]
sources = sys.argv[5:]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    vals = collections.OrderedDict()
    disp = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Dispatch_Id"]
        d = disp.setdefault(k, [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", ""), 0.0])
        d[1] += float(r["Counter_Value"])
    for name, v in disp.values():
        vals.setdefault(name, []).append(v)
    return vals


f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
kernels = {}
for name in f:
    fk, wk = statistics.median(f[name]), statistics.median(w.get(name, [0.0]))
    kernels[name] = {"launches": len(f[name]), "fetch_bytes_per_launch": int(2 * fk * 1024), "write_bytes_per_launch": int(wk * 1024),
                     "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
                     "hbm_bytes_all_launches": int((2 * sum(f[name]) + sum(w.get(name, [0.0]))) * 1024)}
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = ""
res = {"note": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; bytes per launch = (2 x FETCH_SIZE KB [gfx950 correction] + WRITE_SIZE KB) x 1024, "
               "median over the kernel's launches", "command": command, "git_head_at_collection": head,
       "source_sha256": {os.path.basename(s): hashlib.sha256(open(os.path.join(ROOT, "cosyvoice_amd", "csrc", os.path.basename(s)), "rb").read()).hexdigest()
                         for s in sources},
       "units": int(os.environ.get("PMC_UNITS", "0")) or None,
       "total_hbm_bytes": int(sum((2 * sum(f[n]) + sum(w.get(n, [0.0]))) * 1024 for n in f)),
       "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}
json.dump(res, open(out, "w"), indent=1)
for name, k in list(res["kernels"].items())[:10]:
    print(f'{k["launches"]:6d} x {k["hbm_bytes_per_launch"] / 1e6:9.2f} MB (read {k["fetch_bytes_per_launch"] / 1e6:.2f} + written {k["write_bytes_per_launch"] / 1e6:.2f})  {name[:90]}')
