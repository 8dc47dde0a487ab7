"""GPU idle time inside ONE steady-state predict pass, from rocprofv3 --kernel-trace (+ --memory-copy-trace) CSVs.
usage: python tools/step_gaps.py <dir with *_kernel_trace.csv [and *_memory_copy_trace.csv]> [min_gap_us]
A pass = from one `conv_smallcin_u8_kernel<false...` (the detector's first kernel of a call... the first kernel after a gap > 150 us that
follows a softmax/ctc kernel) to the next.  Prints: wall span of the pass, union of kernel-busy time over all streams, and every idle gap
>= min_gap_us with the kernels on both sides."""
import csv, glob, sys
d = sys.argv[1]
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kf)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], "k"))
mf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
for f in mf:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Kind", "?")), "c"))
ev.sort()
# pass boundaries: the rec_resize / first detector stem after a long stretch; use ctc_combine as the end-of-recognition marker
ends = [i for i, e in enumerate(ev) if "ctc_combine" in e[2]]
# group consecutive ctc_combine markers of one pass: a pass ends at the LAST ctc_combine before a detector stem kernel
starts = [i for i, e in enumerate(ev) if "conv_smallcin_u8_kernel<false" in e[2]]
bounds = []
prev = None
for i in starts:
    if prev is None or any(prev < j < i for j in ends):
        bounds.append(i)
    prev = i
if len(bounds) < 4:
    print("not enough passes", len(bounds)); sys.exit(0)
a, b = bounds[-3], bounds[-2]
seg = ev[a:b]
t0, t1 = seg[0][0], max(e[1] for e in seg)
print(f"pass: {len(seg)} events, span {(t1 - t0) / 1e6:.3f} ms (first event -> last end; next pass starts {(ev[b][0] - t0) / 1e6:.3f} ms after)")
busy_k = 0; cur_s = cur_e = None; gaps = []
ks = [e for e in seg if e[3] == "k"]
last = None
for s, e, n, _ in ks:
    if cur_e is None: cur_s, cur_e, last = s, e, n; continue
    if s > cur_e:
        busy_k += cur_e - cur_s
        if (s - cur_e) / 1e3 >= ming: gaps.append(((s - cur_e) / 1e3, (cur_e - t0) / 1e6, last, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last = n if e >= cur_e else last
busy_k += cur_e - cur_s
print(f"kernel-busy (union over streams) {busy_k / 1e6:.3f} ms; sum of kernel durations {sum(e - s for s, e, _, _ in ks) / 1e6:.3f} ms; idle inside the pass {(t1 - t0 - busy_k) / 1e6:.3f} ms; idle before the next pass {(ev[b][0] - t1) / 1e6:.3f} ms")
small = (t1 - t0 - busy_k) / 1e3 - sum(g[0] for g in gaps)
print(f"gaps >= {ming} us: {len(gaps)} totalling {sum(g[0] for g in gaps) / 1e3:.3f} ms; smaller gaps total {small / 1e3:.3f} ms")
for g, at, p, n in gaps:
    print(f"  at {at:7.3f} ms  gap {g:7.1f} us   after {p:50s} before {n}")
cs = [e for e in seg if e[3] == "c"]
if cs:
    print(f"copies in the pass: {len(cs)}, total {sum(e - s for s, e, _, _ in cs) / 1e6:.3f} ms")
