"""Idle time of the queue in front of each kernel, from a rocprofv3 --kernel-trace CSV.

usage: python tools/trace_gaps.py <kernel_trace.csv> [tail_fraction]
Groups the gap (this kernel's start - the previous kernel's end) by the kernel that follows it; gaps above
200 us are host phases and are listed separately."""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
rows = rows[int(len(rows) * (1.0 - tail)):]
by = collections.defaultdict(list)
dur = collections.defaultdict(float)
host = []
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:64]
    if prev_end is not None:
        g = s - prev_end
        if g >= 200000:
            host.append(g / 1e3)
        elif g >= 0:
            by[name].append(g / 1e3)
    dur[name] += (e - s) / 1e3
    prev_end = max(prev_end or 0, e)
tot_k = sum(dur.values())
tot_g = sum(sum(v) for v in by.values())
print(f"kernels {len(rows)}  kernel time {tot_k/1e3:.2f} ms  short gaps {tot_g/1e3:.2f} ms  host gaps {sum(host)/1e3:.2f} ms ({len(host)})")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{k:66s} n={len(v):5d} gap med={statistics.median(v):6.2f} mean={sum(v)/len(v):6.2f} us  kernel avg={dur[k]/max(len(v),1):7.1f} us")
