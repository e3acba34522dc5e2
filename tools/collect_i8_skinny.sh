# rocprofv3 statistics and fabric-traffic counters of the int8 small-M streaming kernel at 28672 x 8192, M = 1 and 16.
# usage (on the GPU box): bash tools/collect_i8_skinny.sh OUTDIR
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/i8_skinny}
mkdir -p $O
for m in 1 16; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_m$m -- python3 tools/pmc_probe.py --a i8 --M $m --N 28672 --K 8192 --launches 500 > $O/stats_m$m.log 2>&1
  cp $(find $O/stats_m$m -name "*kernel_stats.csv" | head -1) $O/m${m}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_m${m}_$c -- python3 tools/pmc_probe.py --a i8 --M $m --N 28672 --K 8192 --launches 50 > $O/pmc_m${m}_$c.log 2>&1
  done
  python3 tools/pmc_summary.py $O/m${m}_pmc.json wq_gemm_i8_skinny $(find $O/pmc_m${m}_* -name "*counter_collection.csv") > /dev/null
done
head -3 $O/m1_kernel_stats.csv $O/m16_kernel_stats.csv
cat $O/m1_pmc.json
