# second, larger campaign (other seeds): same three slices as tools/fuzz_campaign.sh
T=${1:-700}
mkdir -p gpurun_out/s4
export OMP_NUM_THREADS=16
timeout $T python tools/parity_fuzz.py 3000 ${SEED0:-5101} > gpurun_out/s4/fuzz2_plain.log 2>&1 &
timeout $T python tools/parity_fuzz.py 2000 $((${SEED0:-5101}+1)) stages > gpurun_out/s4/fuzz2_stages.log 2>&1 &
timeout $T python tools/parity_fuzz.py 400 $((${SEED0:-5101}+2)) seal > gpurun_out/s4/fuzz2_seal.log 2>&1 &
wait
for f in plain stages seal; do echo "== $f: $(grep -c ' ok$' gpurun_out/s4/fuzz2_$f.log) ok, $(grep -c 'FAIL$' gpurun_out/s4/fuzz2_$f.log) FAIL, $(grep -c 'tolerated' gpurun_out/s4/fuzz2_$f.log) tolerated-rectified"; grep -B1 "FAIL$" gpurun_out/s4/fuzz2_$f.log | cut -c1-300 | head -12; done
