# A/B of run-time knobs on the bench workload: alternating runs on one box.  usage: bash tools/ab_knobs.sh  (writes gpurun_out/s4/ab_*.json)
mkdir -p gpurun_out/s4
B="python bench.py --cpu-pages 0 --no-pipelined --steps 30"
run() { name=$1; shift; env "$@" $B > gpurun_out/s4/ab_$name.json 2> gpurun_out/s4/ab_$name.err; python - <<P
import json
d=json.load(open("gpurun_out/s4/ab_$name.json"))
print("$name", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "us", d["roofline"]["avg_launch_us"], "devres", (d.get("device_resident") or {}).get("value"))
P
}
for i in 1 2 3; do
run base$i X=1
run bands3_$i OAR_BANDS_MULT=3
done
OAR_TIMING=2 python tools/host_entry_breakdown.py 2>&1 | grep -E "subbatch" | tail -5
