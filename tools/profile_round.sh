#!/bin/bash
# One profiling session of the bench workload on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/prof_<tag>/<tag>_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE     -> gpurun_out/pmc_FETCH_SIZE, pmc_WRITE_SIZE   (separate passes: TCC has 4 slots)
#   3. rocprofv3 --pmc <matrix-pipe counters>      -> gpurun_out/pmc_MFMA   (SQ_VALU_MFMA_BUSY_CYCLES, MOPS, GRBM_GUI_ACTIVE, ...)
#   4. the same matrix-pipe counters over tools/mfma_peak (a kernel that IS at the bf16 / f32 MFMA peak): calibrates the
#      utilisation formula used by tools/pmc_summary.py
# PMC passes never carry --kernel-trace / --stats (gpurun refuses the combination).  Then: python tools/make_profile_summaries.py <tag> auto profiles/r5   (auto: the pass count the stats run's bench line reports)
TAG=${1:-r2a}
CFG=${2:-1}          # bench.py --config of the session; != 1: raw passes go to gpurun_out/pmc_*_c<CFG>
SFX=""; if [ "$CFG" != "1" ]; then SFX="_c$CFG"; fi
R=$PWD; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --config $CFG --cpu-pages 0 --no-device-resident --no-pipelined --no-real-size"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- $BENCH --steps 5 --warmup 2 > $R/gpurun_out/prof_$TAG.log 2>&1 < /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c$SFX
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c$SFX -o p -- $BENCH --steps 2 --warmup 1 --no-prof > $R/gpurun_out/pmc_$c$SFX.log 2>&1 < /dev/null
done
MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
rm -rf $R/gpurun_out/pmc_MFMA$SFX $R/gpurun_out/pmc_MFMA_peak
timeout -k 10 500 rocprofv3 --pmc $MFMA --output-format csv -d $R/gpurun_out/pmc_MFMA$SFX -o p -- $BENCH --steps 2 --warmup 1 --no-prof > $R/gpurun_out/pmc_MFMA$SFX.log 2>&1 < /dev/null
if [ ! -x $R/tools/mfma_peak ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/mfma_peak $R/tools/mfma_peak.hip > /dev/null 2>&1; fi
timeout -k 10 200 rocprofv3 --pmc $MFMA --output-format csv -d $R/gpurun_out/pmc_MFMA_peak -o p -- $R/tools/mfma_peak > $R/gpurun_out/pmc_MFMA_peak.log 2>&1 < /dev/null
ls $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_MFMA$SFX $R/gpurun_out/pmc_MFMA_peak | head -20; tail -2 $R/gpurun_out/prof_$TAG.log | cut -c1-300
