#!/bin/bash
# A/B of the host-side DB post-processing on the GPU box: OAR_HOST_FAST=0 (round-4 route) against the default, floating and pinned to
# 2 / 4 / 8 cores (what a rank of bench.py --gpus N gets), alternating so that box drift hits both arms.  Output: gpurun_out/host_ab.txt
mkdir -p gpurun_out/hab
ARGS="--cpu-pages 0 --no-pipelined --no-real-size --no-device-resident --no-prof --steps 20"
show() { python - <<P
import json
try:
    d=json.loads(open("gpurun_out/hab/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], "host cpu ms/step", d["config"].get("host_cpu_ms_per_step"))
except Exception as e: print("$1 failed", e)
P
}
REPS=${1:-2}
for i in $(seq 1 $REPS); do
  for c in free 8 4 2; do
    for f in 0 1; do
      if [ $c = free ]; then OAR_HOST_FAST=$f python bench.py $ARGS > gpurun_out/hab/${c}_fast${f}_$i.json 2>/dev/null
      else OAR_HOST_FAST=$f taskset -c 0-$((c-1)) python bench.py $ARGS > gpurun_out/hab/${c}_fast${f}_$i.json 2>/dev/null; fi
      show ${c}_fast${f}_$i
    done
  done
done 2>&1 | tee gpurun_out/host_ab.txt
