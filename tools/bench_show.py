"""Prints the interesting fields of a bench.py JSON line (file argument): value, roofline by family, extra figures."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "| device_resident", (d.get("device_resident") or {}).get("value"), "| pipelined", (d.get("pipelined") or {}).get("value"),
      "| lighter graphs (r1-r5)", {k: v for k, v in (d.get("lighter_graphs_r1_r5") or d.get("det_real_size") or {}).items() if k != "what"})
r = d.get("roofline") or {}
print("roofline", {k: r.get(k) for k in ("kernel", "bound", "frac", "avg_launch_us", "share_of_step", "traffic")})
for k, v in (r.get("by_family") or {}).items():
    print(f"  {k:18s} bound={v['bound']:4s} frac={v['frac']:.3f} hbm={v['frac_hbm']:.3f} mfma={v['frac_mfma']:.3f} avg_us={v['avg_launch_us']:7.2f} launches/step={v['launches_per_step']:3d} share={v['share_of_step']:.3f}")
print("kernel ms/step", d.get("kernel_ms_per_step_untimed_pass"))
c = d.get("cpu_baseline") or {}
print("cpu_baseline", {k: c.get(k) for k in ("value", "unit", "cores", "kind")})
