#!/usr/bin/env python3
"""The benchmark GEMM (fp16 x int4, 4096^3) in two formulations on the same kernel and the same inputs: the per-channel
scale applied to the weights inside the loop (the reference's fpAintB semantics: half(q) * scale rounded per weight), and
no weight scale in the loop with the scale applied by the Multiply epilogue (the formulation of the reference's
"multiply" multi-ABD instances).  Instruction counts are identical (the unit scale is still multiplied in); what changes
is what the matrix core is fed -- small exact integers instead of full-mantissa products.

    python tools/fast_probe.py
"""
import sys, os, statistics, torch
sys.path.insert(0, os.getcwd())
from composable_kernel_amd import capi
import bench
dev = torch.device("cuda", 0)
M = N = K = 4096
a, q, s, bp = bench.synth_inputs(M, N, K, 11939, dev)
c = torch.empty((M, N), dtype=torch.float16, device=dev)
sp = torch.cuda.current_stream().cuda_stream
p0 = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K)
p1 = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), None, c.data_ptr(), M, N, K, epilogue=capi.EPI_MULTIPLY, d0=s.data_ptr())
for name, p in (("scale in loop", p0), ("unit scale in loop, scale in epilogue", p1), ("scale in loop", p0), ("unit scale in loop, scale in epilogue", p1)):
    capi.run_timed(p, None, sp, 3, 2000)
    ts = [capi.run_timed(p, None, sp, 0, 50) for _ in range(7)]
    ms = statistics.median(ts)
    print(name, capi.kernel_name(p), round(ms * 1e3, 2), "us", round(2.0 * M * N * K / ms / 1e9, 1), "TFLOP/s", flush=True)
