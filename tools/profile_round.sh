R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1j -o r1j -- python $R/bench.py --steps 5 --warmup 2 --cpu-pages 0 > $R/gpurun_out/prof_r1j.log 2>&1 < /dev/null
for c in FETCH_SIZE WRITE_SIZE SQ_WAVES; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --cpu-pages 0 --no-prof > $R/gpurun_out/pmc_$c.log 2>&1 < /dev/null
done
ls $R/gpurun_out/prof_r1j $R/gpurun_out/pmc_FETCH_SIZE | head; tail -2 $R/gpurun_out/prof_r1j.log | cut -c1-400
cd $R; timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_final2.json 2> gpurun_out/bench_final2.err; tail -c 600 gpurun_out/bench_final2.json
