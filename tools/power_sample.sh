# Sample the GPU's socket power and engine clock while a fixed loop of one kernel runs (read-only amd-smi / rocm-smi queries).
# usage (on the GPU box): bash tools/power_sample.sh OUT.txt LABEL -- python3 tools/f8_bench_loop.py 60000 16384
set -e
cd $GRAFT_REPO_ROOT
OUT=$1; LABEL=$2; shift 3
"$@" > $OUT.run 2>&1 &
PID=$!
sleep 6   # import + set-up + clock settling
echo "== $LABEL" >> $OUT
for i in 1 2 3 4 5; do
  if kill -0 $PID 2>/dev/null; then
    P=$(/opt/rocm/bin/amd-smi metric -p 2>/dev/null | grep SOCKET_POWER | awk '{print $2}')
    C=$(/opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep "sclk clock level" | sed 's/.*(\(.*\)Mhz).*/\1/')
    echo "socket_power_w $P sclk_mhz $C" >> $OUT
    sleep 0.4
  fi
done
wait $PID
tail -1 $OUT.run | cut -c1-160 >> $OUT
