#!/bin/bash
# one gpurun call at the end of a round: GPU suite, smoke, two default bench lines (kept under gpurun_out/, copied to profiles/ by hand)
TAG=${1:-r5h}
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/gpu_tests_$TAG.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/gpu_tests_$TAG.txt
for i in 1 2; do python bench.py --steps 20 > gpurun_out/bench_${TAG}_$i.json 2> /dev/null; python tools/bench_show.py gpurun_out/bench_${TAG}_$i.json | head -3; done
