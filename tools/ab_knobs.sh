# where the caller's threads run: floating / taskset from outside / affinity set in-process BEFORE anything touches the HIP runtime
mkdir -p gpurun_out/s4
ARGS="--cpu-pages 0 --no-pipelined --steps 20"
show() { python - <<P
import json
d=json.load(open("gpurun_out/s4/ab_$1.json"))
print("$1", d["value"], d["ms_per_step"], "devres", (d.get("device_resident") or {}).get("value"))
P
}
for i in 1 2 3; do
python bench.py $ARGS > gpurun_out/s4/ab_free_$i.json 2>/dev/null; show free_$i
taskset -c 0-15 python bench.py $ARGS > gpurun_out/s4/ab_taskset_$i.json 2>/dev/null; show taskset_$i
python -c "import os, sys, runpy; os.sched_setaffinity(0, range(16)); sys.argv = ['bench.py'] + '$ARGS'.split(); runpy.run_path('bench.py', run_name='__main__')" > gpurun_out/s4/ab_early_inproc_$i.json 2>/dev/null; show early_inproc_$i
OAR_HOST_THREADS=14 python bench.py $ARGS > gpurun_out/s4/ab_free_ht14_$i.json 2>/dev/null; show free_ht14_$i
done
