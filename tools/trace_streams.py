"""Which kernels ran on which HIP stream / HSA queue, and when: summary of a rocprofv3 --kernel-trace CSV (kernels on the default stream inside a
pipelined run are implicit barriers against the CU-masked streams).  usage: trace_streams.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
by = collections.defaultdict(list)
for r in rows:
    by[(r["Stream_Id"], r["Queue_Id"], r["Thread_Id"])].append(r)
for (sid, qid, tid), rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
    c = collections.Counter(r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:60] for r in rs)
    a, b = min(int(r["Start_Timestamp"]) for r in rs), max(int(r["End_Timestamp"]) for r in rs)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"stream {sid} queue {qid} thread {tid}: {len(rs)} kernels, active {1e-6 * (a - t0):.0f}..{1e-6 * (b - t0):.0f} ms, busy {1e-6 * busy:.0f} ms; "
          + " | ".join(f"{n} x{k}" for n, k in c.most_common(3)))
