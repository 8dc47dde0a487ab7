#!/bin/bash
# knob A/B at 2 / 4 pinned cores and floating: tools/exp_cores.sh "<ENV=VAL>" "<ENV=VAL>" ...   (each argument is one arm; "" = defaults)
mkdir -p gpurun_out/e3
ARGS="--cpu-pages 0 --no-pipelined --no-real-size --no-device-resident --no-prof --steps 20"
show() { python - <<P
import json
try:
    d=json.loads(open("gpurun_out/e3/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], "cpu", d["config"].get("host_cpu_ms_per_step"))
except Exception as e: print("$1 failed", e)
P
}
for i in 1 2; do
  for c in free 4 2; do
    k=0
    for arm in "$@"; do
      k=$((k+1))
      if [ $c = free ]; then env $arm python bench.py $ARGS > gpurun_out/e3/${c}_arm${k}_$i.json 2>/dev/null
      else env $arm taskset -c 0-$((c-1)) python bench.py $ARGS > gpurun_out/e3/${c}_arm${k}_$i.json 2>/dev/null; fi
      echo -n "[$arm] "; show ${c}_arm${k}_$i
    done
  done
done
