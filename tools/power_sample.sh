# Sample the GPU's power and clocks while a fixed loop of one kernel runs (read-only rocm-smi queries).
# usage (on the GPU box): bash tools/power_sample.sh OUT.txt -- python3 tools/f8_bench_loop.py 20000 16384
set -e
cd $GRAFT_REPO_ROOT
OUT=$1; shift 2
"$@" > $OUT.run 2>&1 &
PID=$!
sleep 6   # import + set-up
for i in 1 2 3 4 5 6; do
  if kill -0 $PID 2>/dev/null; then
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" >> $OUT || true
    echo "--" >> $OUT
    sleep 0.5
  fi
done
wait $PID
tail -1 $OUT.run >> $OUT
