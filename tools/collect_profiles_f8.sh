set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2final_f8
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/f8_bench_loop.py 2000 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/f8_bench_loop.py 200 > $O/pmc_$c.log 2>&1
done
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 tools/f8_bench_loop.py 1000 > $O/pmc_sq.log 2>&1
# bf16 x int8 at the FFN shape (config 3), pre-shuffled int8 weights
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 bench.py --config C3 --no-cpu > $O/c3_under_rocprof.json 2> $O/c3.err
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv"
