#!/bin/bash
# rocprofv3 kernel stats + bench lines of BASELINE configs 2 (server graphs, 64 x 1280^2) and 4 (all optional stages) -- VERDICT r3 #4c.
# usage (GPU box, repo root): bash tools/profile_configs.sh <tag>   -> gpurun_out/prof_<tag>_c{2,4}/, gpurun_out/bench_<tag>_c{2,4}.json
TAG=${1:-r4}
R=$PWD; cd /tmp; export TMPDIR=/tmp
for c in 2 4; do
  timeout -k 10 600 python $R/bench.py --config $c --cpu-pages 0 2> /dev/null | tail -1 > $R/gpurun_out/bench_${TAG}_c$c.json
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_c$c -o ${TAG}_c$c -- python $R/bench.py --config $c --cpu-pages 0 --no-device-resident --no-pipelined --steps 3 --warmup 1 > $R/gpurun_out/prof_${TAG}_c$c.log 2>&1 < /dev/null
done
ls $R/gpurun_out/prof_${TAG}_c2 $R/gpurun_out/prof_${TAG}_c4 | head
