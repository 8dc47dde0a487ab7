# host-side phase sequence of the bench workload's predict (OAR_TIMING=2), old and new order of finish(sb - 1); DET_LAST variants
mkdir -p gpurun_out/s4
for v in "OAR_DET_FINISH_EARLY=0" "OAR_DET_FINISH_EARLY=1" "OAR_DET_LAST=2" ; do
  echo "== $v"; env $v OAR_TIMING=2 python tools/host_entry_breakdown.py 2> gpurun_out/s4/timing_$v.txt | tail -1
  grep "ocr.predict" gpurun_out/s4/timing_$v.txt | tail -4
done
