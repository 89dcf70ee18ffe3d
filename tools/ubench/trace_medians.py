"""Median duration / preceding gap per kernel (second half of the run) from a rocprofv3 kernel-trace csv."""
import csv, statistics, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = collections.defaultdict(list); gaps = collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r['Kernel_Name'][:70] + f" g{r.get('Grid_Size_X', r.get('Grid_Size', ''))}"
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    d[name].append(e - s)
    if prev_end is not None: gaps[name].append(s - prev_end)
    prev_end = e
for k, v in d.items():
    v2 = v[len(v) // 2:]; g = gaps[k][len(gaps[k]) // 2:] or [0]
    print(f"{k:82s} n={len(v):5d} median {statistics.median(v2):7.0f} ns  min {min(v2):6d}  gap-before {statistics.median(g):6.0f}")
