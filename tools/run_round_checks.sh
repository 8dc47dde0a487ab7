#!/bin/bash
# scratch driver for one gpurun call (edited per experiment)
ARGS="--cpu-pages 0 --no-pipelined --no-real-size --no-device-resident --no-prof --steps 20"
for i in 1 2 3; do for arm in "OAR_DET_SUB=16" "OAR_DET_SUB=20" "OAR_DET_SUB=24" "OAR_DET_SUB=16 OAR_DET_LAST=6" "OAR_DET_SUB=16 OAR_DET_LAST=12" "OAR_DET_SUB=16 OAR_DET_FIRST=0"; do env $arm python bench.py $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'])"; done; done
for c in 4 2; do for arm in "OAR_DET_SUB=8" "OAR_DET_SUB=16" "OAR_DET_SUB=8" "OAR_DET_SUB=16"; do env $arm taskset -c 0-$((c-1)) python bench.py $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cores $c $arm', d['value'], d['ms_per_step'])"; done; done
