#!/bin/bash
# host border follower vs the GPU one (contours.hip), and the GPU one on CU-masked streams (OAR_CONTOUR_CUS=n, pipeline.cc).
# BASELINE configs[1] (32 x 960^2 pages), synchronous host-entry metric.  usage (GPU box, repo root): bash tools/contours_breakeven.sh > gpurun_out/gpu_contours_cu_mask.txt
run() { python bench.py --cpu-pages 0 --no-device-resident --no-pipelined 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for t in 1 2 4 16; do
  echo "host_threads=$t gpu_contours=0 -> $(OAR_HOST_THREADS=$t OAR_GPU_CONTOURS=0 run)"
  echo "host_threads=$t gpu_contours=1 -> $(OAR_HOST_THREADS=$t OAR_GPU_CONTOURS=1 run)"
  for n in 8 32 128; do
    echo "host_threads=$t gpu_contours=1 contour_cus=$n -> $(OAR_HOST_THREADS=$t OAR_GPU_CONTOURS=1 OAR_CONTOUR_CUS=$n run)"
  done
done
