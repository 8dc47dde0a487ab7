#!/bin/bash
# scratch driver for one gpurun call (edited per experiment)
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py tests/test_gpu_async.py tests/test_gpu_config5.py -x -q 2>&1 | tail -4
ARGS="--cpu-pages 0 --no-pipelined --no-real-size --no-device-resident --no-prof --steps 20"
for i in 1 2 3 4; do for arm in "X=1" "OAR_REC_BATCH_BACK=0"; do env $arm python bench.py $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'])"; done; done
python tools/host_entry_packed.py
