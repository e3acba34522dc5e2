# Issue-side counters of the fp8 x int4 kernel in three passes of at most 8 SQ counters each.
# usage (on the GPU box): bash tools/collect_f8_sq.sh OUTDIR K LAUNCHES [FLAGS]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$1; K=${2:-4096}; N=${3:-300}; F=${4:-0}
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p1 -- python3 tools/f8_bench_loop.py $N $K $F > $O/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL --output-format csv -d $O/p2 -- python3 tools/f8_bench_loop.py $N $K $F > $O/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_IFETCH --output-format csv -d $O/p3 -- python3 tools/f8_bench_loop.py $N $K $F > $O/p3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/p4 -- python3 tools/f8_bench_loop.py $N $K $F > $O/p4.log 2>&1
python3 tools/pmc_summary.py $O/summary.json wq_gemm_f8_mfma_256 $(find $O -name "*counter_collection.csv")
