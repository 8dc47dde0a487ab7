"""Turns the raw rocprofv3 output of one profiling session (gpurun_out/prof_<tag>/ + gpurun_out/pmc_*/) into the small
files committed under profiles/<round>/:
  <tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary, verbatim
  <tag>_kernel_stats.txt        per-step view of the same file (tools/stats_summary.py)
  <tag>_pmc_summary.txt         FETCH_SIZE (x2, gfx950 correction) / WRITE_SIZE / SQ counters per kernel (tools/pmc_summary.py)
  pmc_traffic.json              HBM bytes per launch for each kernel class, read by bench.py for roofline.traffic
usage: python tools/make_profile_summaries.py <tag> <steps_in_stats_run> [round_dir]
"""
import csv, collections, json, shutil, subprocess, sys
from pathlib import Path

tag, steps = sys.argv[1], sys.argv[2]
out = Path(sys.argv[3] if len(sys.argv) > 3 else "profiles/r1")
out.mkdir(parents=True, exist_ok=True)
src = Path("gpurun_out")
shutil.copy(src / f"prof_{tag}" / f"{tag}_kernel_stats.csv", out / f"{tag}_kernel_stats.csv")
txt = subprocess.run([sys.executable, "tools/stats_summary.py", str(src / f"prof_{tag}" / f"{tag}_kernel_stats.csv"), steps, "40"], capture_output=True, text=True).stdout
(out / f"{tag}_kernel_stats.txt").write_text(txt)
txt = subprocess.run([sys.executable, "tools/pmc_summary.py", str(src), "40"], capture_output=True, text=True).stdout
(out / f"{tag}_pmc_summary.txt").write_text(txt)

def klass(name):   # kernel name -> profiler class used by the in-library profiler / bench.py
    for k in ("conv_igemm_ws_x6", "conv_igemm_ws", "conv_igemm", "conv_dw", "conv_smallcin", "softmax_argmax", "rec_pack", "global_avgpool", "binary", "resize", "copy2d", "normalize", "gemm_batched", "permute"):
        if k in name:
            return k
    return None

def per_class(path, counter):
    tot = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        c = klass(r["Kernel_Name"])
        if c:
            tot[c] += float(r["Counter_Value"]); disp[c].add(r["Dispatch_Id"])
    return {c: (tot[c], len(disp[c])) for c in tot}

f = per_class(src / "pmc_FETCH_SIZE" / "p_counter_collection.csv", "FETCH_SIZE")
w = per_class(src / "pmc_WRITE_SIZE" / "p_counter_collection.csv", "WRITE_SIZE")
res = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 2 --warmup 1 --cpu-pages 0 --no-prof`; "
                "counters are in KiB; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (HBM section); bytes are averages per launch of the class",
       "_source_tag": tag}
for c in f:
    fb = f[c][0] * 1024 * 2 / max(f[c][1], 1)
    wb = w.get(c, (0, 1))[0] * 1024 / max(w.get(c, (0, 1))[1], 1)
    res[c] = {"fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb), "launches_sampled": f[c][1]}
(out / "pmc_traffic.json").write_text(json.dumps(res, indent=1) + "\n")
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1)[:1500])
