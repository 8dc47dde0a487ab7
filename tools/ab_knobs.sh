# A/B of run-time knobs on the bench workload: alternating runs on one box.  usage: bash tools/ab_knobs.sh  (writes gpurun_out/s4/ab_*.json)
mkdir -p gpurun_out/s4
B="python bench.py --cpu-pages 0 --steps 20"
run() { name=$1; shift; env "$@" $B > gpurun_out/s4/ab_$name.json 2> gpurun_out/s4/ab_$name.err; python - <<P
import json
d=json.load(open("gpurun_out/s4/ab_$name.json"))
print("$name", d["value"], d["ms_per_step"], "devres", (d.get("device_resident") or {}).get("value"), "pipelined", (d.get("pipelined") or {}).get("value"))
P
}
for i in 1 2 3 4; do
run new_$i X=1
run nothread_$i OAR_DET_ENQ_THREAD=0
run old_$i OAR_DET_ENQ_THREAD=0 OAR_DET_FINISH_EARLY=0 OAR_FINISH_CHUNKS=0
done
