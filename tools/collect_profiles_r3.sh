# Kernel-trace statistics (warm, >= 1000 launches each) and counter passes for the four matrix-core kernels at their
# BASELINE shapes.  usage (on the GPU box): bash tools/collect_profiles_r3.sh OUTDIR
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/r3final}
mkdir -p $O
SQ="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
# C2: bench.py itself
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --no-cpu > $O/c2_bench_under_rocprof.json 2> $O/c2.err
# C3: bench.py --config C3
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 bench.py --config C3 --no-cpu > $O/c3_bench_under_rocprof.json 2> $O/c3.err
# C5: fixed loop of the fp8 x int4 kernel
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_f8 -- python3 tools/f8_bench_loop.py 3000 > $O/f8_stats.log 2>&1
# C1 at 4096^3: fixed loop of the int8 x int8 256-tile kernel
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_i8 -- python3 tools/pmc_probe.py --a i8 --M 4096 --N 4096 --K 4096 --launches 3000 > $O/i8_stats.log 2>&1
pass() { # name counters... -- program args
  local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -s KILL 120 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $O/pmc_$name -- "$@" > $O/pmc_$name.log 2>&1
}
for c in FETCH_SIZE WRITE_SIZE; do
  pass c2_$c $c -- python3 tools/pmc_probe.py --a f16 --b i4pre --M 4096 --N 4096 --K 4096 --launches 100
  pass c3_$c $c -- python3 tools/pmc_probe.py --a bf16 --b i8pre --M 4096 --N 11008 --K 4096 --launches 100
  pass f8_$c $c -- python3 tools/f8_bench_loop.py 200
  pass i8_$c $c -- python3 tools/pmc_probe.py --a i8 --M 4096 --N 4096 --K 4096 --launches 100
done
pass c2_sq $SQ -- python3 tools/pmc_probe.py --a f16 --b i4pre --M 4096 --N 4096 --K 4096 --launches 300
pass c3_sq $SQ -- python3 tools/pmc_probe.py --a bf16 --b i8pre --M 4096 --N 11008 --K 4096 --launches 300
pass f8_sq $SQ -- python3 tools/f8_bench_loop.py 1000
pass i8_sq $SQ -- python3 tools/pmc_probe.py --a i8 --M 4096 --N 4096 --K 4096 --launches 1000
python3 tools/pmc_summary.py $O/c2_pmc.json wq_gemm_mfma_256 $(find $O/pmc_c2_* -name "*counter_collection.csv") > /dev/null
python3 tools/pmc_summary.py $O/c3_pmc.json wq_gemm_mfma_256 $(find $O/pmc_c3_* -name "*counter_collection.csv") > /dev/null
python3 tools/pmc_summary.py $O/f8_pmc.json wq_gemm_f8_mfma_256 $(find $O/pmc_f8_* -name "*counter_collection.csv") > /dev/null
python3 tools/pmc_summary.py $O/i8_pmc.json wq_gemm_i8_mfma_256 $(find $O/pmc_i8_* -name "*counter_collection.csv") > /dev/null
for k in c2 c3 f8 i8; do cp $(find $O/stats_$k -name "*kernel_stats.csv" | head -1) $O/${k}_kernel_stats.csv; done
timeout -k 10 150 python bench.py > $O/bench.json 2> $O/bench.err
for c in C1 C3 C5; do timeout -k 10 150 python bench.py --config $c >> $O/bench_per_config.jsonl 2>> $O/bench_per_config.err; done
timeout -k 10 200 python bench.py --config C4 --gpus 1 > $O/splitk_world1.json 2> $O/splitk_world1.err
head -3 $O/*_kernel_stats.csv
cut -c1-300 $O/bench.json
