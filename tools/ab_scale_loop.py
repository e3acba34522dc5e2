#!/usr/bin/env python3
"""A short fixed loop of the block-scaled fp8 x int4 GEMM (4096^3, bf16 out) for profiler runs.

    python3 tools/ab_scale_loop.py [launches] [scale_block_m: 128 | 1]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402

M = N = K = 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bm = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
sp = torch.cuda.current_stream().cuda_stream
a = (torch.randint(24, 88, (M, K), generator=g, device=dev, dtype=torch.int16) +
     128 * torch.randint(0, 2, (M, K), generator=g, device=dev, dtype=torch.int16)).to(torch.uint8)
q = torch.randint(-8, 8, (N, K), generator=g, device=dev, dtype=torch.int8)
b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
b2 = torch.empty_like(b)
capi.preshuffle_int4(b.data_ptr(), K // 2, K, N, b2.data_ptr(), K, sp)
sa = torch.rand(((M + bm - 1) // bm, K // 128), generator=g, device=dev)
sb = torch.rand((N // 128, K // 128), generator=g, device=dev)
c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
p = capi.fpAintB_problem(a.data_ptr(), b2.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_E4M3FN, b_dtype=capi.I4_PRESHUFFLED,
                         c_dtype=capi.BF16, a_block_scale=sa.data_ptr(), b_block_scale=sb.data_ptr(), scale_block=(bm, 128, 128))
assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
for _ in range(n):
    capi.launch(p, None, sp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    capi.launch(p, None, sp)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
print(capi.kernel_name(p), 2 * n, f"launches; second half: {us:.1f} us per launch = {2.0 * M * N * K / us * 1e-6:.0f} TFLOP/s")
