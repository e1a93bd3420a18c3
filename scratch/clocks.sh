#!/bin/bash
# shader clock / power while the bench runs: scratch/clocks.sh [bench flags]   -> gpurun_out/clocks_<tag>.txt
TAG=${TAG:-run}
OUT=gpurun_out/clocks_$TAG.txt
rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -E "sclk|Power|power" > $OUT
echo "--- under load: python bench.py $@" >> $OUT
python bench.py --steps ${STEPS:-4000} --warmup 4 --no-cpu-baseline --pipelined 0 --profile-steps 0 "$@" > /tmp/bench_clk.json 2>/dev/null &
PID=$!
sleep ${DELAY:-9}
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket" | tr '\n' ' ' >> $OUT; echo >> $OUT
  sleep 0.3
done
wait $PID
python -c "
import json; d=json.loads([l for l in open('/tmp/bench_clk.json') if l.startswith('{')][0]); print('value', round(d['value'],1), d['unit'])" >> $OUT
cat $OUT
