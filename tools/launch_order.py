"""Ordered launch list of ONE warm graph run from a rocprofv3 --kernel-trace csv.
usage (GPU box):  cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lo_rec -- python $R/tools/rec_trace.py rec
                  python tools/launch_order.py gpurun_out/lo_rec   -> the launches of the LAST run (rec_trace.py runs the graph twice), name / duration / gap before"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 2
rows = rows[len(rows) - n:]
prev = None
tot = 0
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    tot += e - s
    print(f"{i:3d} {r['Kernel_Name'][:90]:90s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')} wg {r.get('Workgroup_Size_X', '?')}")
print("launches", len(rows), "kernel us", round(tot / 1e3, 1), "span us", round((int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3, 1))
