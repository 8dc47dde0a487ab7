ARGS="--cpu-pages 0 --no-pipelined --no-real-size --no-device-resident --no-prof --steps 20"
for i in 1 2 3 4; do for arm in "X=1" "OAR_UPLOAD_THREADS=2"; do env $arm python bench.py $ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'])"; done; done
