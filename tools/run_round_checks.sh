python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -5 > gpurun_out/gputest_r5h.txt; cat gpurun_out/gputest_r5h.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for w in rec det; do rm -rf $R/gpurun_out/lo_$w; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lo_$w -- python $R/tools/rec_trace.py $w > /dev/null 2>&1; python $R/tools/launch_order.py $R/gpurun_out/lo_$w > $R/gpurun_out/lo_$w.txt 2>&1; rm -rf $R/gpurun_out/lo_$w; done
cd $R
ARGS="--cpu-pages 0 --no-pipelined --no-real-size --no-device-resident --no-prof --steps 20"
for i in 1 2 3; do for arm in "X=1" "OAR_FUSE_SE_POOL=1" "OAR_DB_FINISH_FUSED=0"; do env $arm python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'])"; done; done 2>&1 | tee gpurun_out/ab_r5h.txt
