#!/usr/bin/env python3
"""A short fixed loop of the fp8 x int4 GEMM (4096^3, MultiplyMultiply + FastGelu) for profiler runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402

M = N = 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
a = (torch.randint(24, 88, (M, K), generator=g, device=dev, dtype=torch.int16) +
     128 * torch.randint(0, 2, (M, K), generator=g, device=dev, dtype=torch.int16)).to(torch.uint8)
q = torch.randint(-8, 8, (N, K), generator=g, device=dev, dtype=torch.int8)
b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
sp = torch.cuda.current_stream().cuda_stream
capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
bdt = capi.I4_PACKED
if os.environ.get("WQ_PRESHUFFLE", "1") != "0":  # the weights in the pre-shuffled order (the deployed form)
    b2 = torch.empty_like(b)
    capi.preshuffle_int4(b.data_ptr(), K // 2, K, N, b2.data_ptr(), K, sp)
    b, bdt = b2, capi.I4_PRESHUFFLED
d0 = torch.rand((N,), generator=g, device=dev)
d1 = torch.rand((M,), generator=g, device=dev)
c = torch.empty((M, N), dtype=torch.float16, device=dev)
p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_E4M3FN, b_dtype=bdt, c_dtype=capi.F16,
                         epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU, d0=d0.data_ptr(), d1=d1.data_ptr(),
                         flags=int(sys.argv[3]) if len(sys.argv) > 3 else 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for _ in range(n):
    capi.launch(p, None, sp)
torch.cuda.synchronize()
if os.environ.get("WQ_TIME"):   # a second, timed pass on the now warm chip
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        capi.launch(p, None, sp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"second pass: {us:.1f} us per launch = {2.0 * M * N * K / us * 1e-6:.0f} TFLOP/s")
print(capi.kernel_name(p), n, "launches")
