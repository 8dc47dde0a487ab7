# Long randomised parity campaign on one GPU box: three slices of tools/parity_fuzz.py side by side (fresh seeds), logs under gpurun_out/s4/.
# usage: bash tools/fuzz_campaign.sh [seconds per slice]
T=${1:-1300}
mkdir -p gpurun_out/s4
export OMP_NUM_THREADS=16
timeout $T python tools/parity_fuzz.py 400 ${SEED0:-4101} > gpurun_out/s4/fuzz_plain.log 2>&1 &
timeout $T python tools/parity_fuzz.py 200 $((${SEED0:-4101}+1)) stages > gpurun_out/s4/fuzz_stages.log 2>&1 &
timeout $T python tools/parity_fuzz.py 150 $((${SEED0:-4101}+2)) seal > gpurun_out/s4/fuzz_seal.log 2>&1 &
wait
for f in plain stages seal; do echo "== $f: $(grep -c ' ok$' gpurun_out/s4/fuzz_$f.log) ok, $(grep -c 'FAIL$' gpurun_out/s4/fuzz_$f.log) FAIL, $(grep -c 'tolerated' gpurun_out/s4/fuzz_$f.log) tolerated-rectified"; tail -1 gpurun_out/s4/fuzz_$f.log; done
