#!/usr/bin/env python3
"""Timing of fp8 x int4 kernel variants (flags values on the command line) at 4096^3; results are not checked."""
import statistics, sys
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from composable_kernel_amd import capi
import os
M = N = 4096
K = int(os.environ.get('K', 4096))
dev = torch.device("cuda", 0)
sp = torch.cuda.current_stream().cuda_stream
a = torch.randint(0, 0x50, (M, K), device=dev, dtype=torch.int16).to(torch.uint8)
q = torch.randint(-8, 8, (N, K), device=dev, dtype=torch.int8)
b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
d0 = torch.rand((N,), device=dev); d1 = torch.rand((M,), device=dev)
c = torch.zeros((M, N), dtype=torch.float16, device=dev)
for f in [int(x) for x in sys.argv[1:]]:
    p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16,
                             epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU, d0=d0.data_ptr(), d1=d1.data_ptr(), flags=f)
    capi.run_timed(p, None, sp, 0, 300)
    ts = [capi.run_timed(p, None, sp, 2, 100) for _ in range(5)]
    ms = statistics.median(ts)
    print(f"flags {f}: {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
