set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2final
mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/pmc_probe.py --a f16 --b i4pre --M 4096 --N 4096 --K 4096 --launches 100 > $O/pmc_$c.log 2>&1
done
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 tools/pmc_probe.py --a f16 --b i4pre --M 4096 --N 4096 --K 4096 --launches 100 > $O/pmc_sq.log 2>&1
timeout -k 10 150 python bench.py > $O/bench.json 2> $O/bench.err
for c in C1 C3 C5; do timeout -k 10 150 python bench.py --config $c >> $O/bench_per_config.jsonl 2>> $O/bench_per_config.err; done
timeout -k 10 200 python tools/config_bench.py > $O/configs.jsonl 2> $O/configs.err || true
find $O -name "*.csv" | head -20
cat $O/bench.json | cut -c1-400
