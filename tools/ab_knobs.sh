# A/B of run-time knobs on the bench workload: alternating runs on one box.  usage: bash tools/ab_knobs.sh  (writes gpurun_out/s4/ab_*.json)
mkdir -p gpurun_out/s4
B="python bench.py --cpu-pages 0 --no-pipelined --steps 30"
run() { name=$1; shift; $B "$@" > gpurun_out/s4/ab_$name.json 2> gpurun_out/s4/ab_$name.err; python - <<P
import json
d=json.load(open("gpurun_out/s4/ab_$name.json"))
print("$name", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "us", d["roofline"]["avg_launch_us"], "devres", (d.get("device_resident") or {}).get("value"), "kern", sum((d.get("kernel_ms_per_step_untimed_pass") or {}).values()))
P
}
for i in 1 2; do
for rb in 256 128 192 320 384 512; do
run rb${rb}_$i --region-batch $rb
done
done
