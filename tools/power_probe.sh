# Samples the card's power draw and shader clock while the replayed training step runs (a few seconds of steps):
# tells whether the step runs against the power limit (clock below its maximum at the limit) or not.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/power; mkdir -p $O
cd $R
(timeout -k 5 120 python bench.py --no-cpu-baseline --single-dtype --no-extra --steps 600 --warmup 20 > $O/bench_long.json 2> $O/bench_long.err; echo BENCH_DONE >> $O/samples.txt) &
BP=$!
: > $O/samples.txt
for i in $(seq 1 150); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo "t=$(date +%s.%N)" >> $O/samples.txt
  timeout 5 rocm-smi --showpower --showclocks --showuse 2>&1 | grep -E "Power|sclk|mclk|GPU use" >> $O/samples.txt
  sleep 0.15
done
wait $BP
timeout 5 rocm-smi --showmaxpower --showpower 2>&1 | grep -E "Power|Max" >> $O/samples.txt
echo PROBE_DONE
