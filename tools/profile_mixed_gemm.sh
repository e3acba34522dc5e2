#!/bin/bash
# Shape sweep of the weight-quantized GEMM profiler, with the argument order and the shape list of the reference's
# script/profile_mixed_gemm.sh:17-51 (M, N in {16, 2048, 8192} x K in {1024, 8192, 65536}; strides -1 = packed).
#   tools/profile_mixed_gemm.sh gemm_splitk 0 1 0 2 0 1 4      # op datatype layout verify init log time KBatch
export HIP_VISIBLE_DEVICES=0
DRIVER="$(dirname "$0")/../examples/example_wq_profiler"
OP=$1; DATATYPE=$2; LAYOUT=$3; VERIFY=$4; INIT=$5; LOG=$6; TIME=$7; KBatch=$8
for M in 16 2048 8192; do
  for N in 16 2048 8192; do
    for K in 1024 8192 65536; do
      if [ "$OP" = "gemm_splitk" ]; then
        $DRIVER $OP $DATATYPE $LAYOUT $VERIFY $INIT $LOG $TIME $M $N $K -1 -1 -1 $KBatch
      else
        $DRIVER $OP $DATATYPE $LAYOUT $VERIFY $INIT $LOG $TIME $M $N $K -1 -1 -1
      fi
    done
  done
done
