# rocprofv3 statistics and counters of the two block-scaled fp8 x int4 kernels at 4096^3 (128^3 blocks; 1 x 128 activation scales).
# usage (on the GPU box): bash tools/collect_abscale.sh OUTDIR
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/abscale}
mkdir -p $O
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
for bm in 128 1; do
  name=abscale; [ $bm = 1 ] && name=abscale_rows
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- python3 tools/ab_scale_loop.py 1000 $bm > $O/stats_$name.log 2>&1
  (head -1 $(find $O/stats_$name -name "*kernel_stats.csv" | head -1); grep '"wq::\|"void wq::' $(find $O/stats_$name -name "*kernel_stats.csv" | head -1)) > $O/${name}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${name}_$c -- python3 tools/ab_scale_loop.py 50 $bm > $O/pmc_${name}_$c.log 2>&1
  done
  timeout -s KILL 120 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_${name}_sq -- python3 tools/ab_scale_loop.py 150 $bm > $O/pmc_${name}_sq.log 2>&1
  python3 tools/pmc_summary.py $O/${name}_pmc.json wq_gemm_f8_mfma_256_$name $(find $O/pmc_${name}_* -name "*counter_collection.csv") > /dev/null
  cat $O/${name}_kernel_stats.csv
done
