"""Per-step view of a rocprofv3 --kernel-trace --stats summary (kernel_stats.csv).
usage: python tools/stats_summary.py <kernel_stats.csv> <passes | log:<bench log>> [rows]
The divisor is the number of oar_ocr_predict passes the profiled command made -- ALL of them: bench.py's two untimed discovery passes
and its warm-up passes run the same kernels as the timed steps.  `log:<file>` reads it from the bench line in that log
("predict_passes_total"); round 2 divided by the timed steps only, which overstated every per-step column 1.8x."""
import csv, json, sys

path, arg = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "7"
if arg.startswith("log:"):
    passes = None
    for line in open(arg[4:], errors="replace"):
        if line.startswith("{") and "predict_passes_total" in line:
            passes = float(json.loads(line)["predict_passes_total"])
    if not passes:
        sys.exit(f"no bench line with predict_passes_total in {arg[4:]}")
else:
    passes = float(arg)
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"predict passes in the profiled run: {passes:g}")
print("total kernel ms/pass", round(tot / passes / 1e6, 2))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 26]:
    print("%-72s calls/pass=%6.1f ms/pass=%6.2f avg_us=%7.1f" % (r["Name"][:72], int(r["Calls"]) / passes, float(r["TotalDurationNs"]) / passes / 1e6, float(r["AverageNs"]) / 1e3))
