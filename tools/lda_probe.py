#!/usr/bin/env python3
"""Does the row pitch of the activations matter?  fp16 x int4 (pre-shuffled) 4096^3 with lda = K and lda = K + pad."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402

M = N = K = 4096
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1)
sp = torch.cuda.current_stream().cuda_stream
q = torch.randint(-8, 8, (N, K), generator=g, device=dev, dtype=torch.int8)
b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
b2 = torch.empty_like(b)
capi.preshuffle_int4(b.data_ptr(), K // 2, K, N, b2.data_ptr(), K, sp)
s = (torch.rand((N,), generator=g, device=dev) * 0.05 + 0.004).half()
probs, keep = {}, []
for pad in (0, 64, 128, 512):
    a = (torch.rand((M, K + pad), generator=g, device=dev) * 2 - 1).half()
    for cpad in (0, 64):
        c = torch.empty((M, N + cpad), dtype=torch.float16, device=dev)
        keep += [a, c]
        probs[(pad, cpad)] = capi.fpAintB_problem(a.data_ptr(), b2.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, lda=K + pad,
                                                  ldc=N + cpad, b_dtype=capi.I4_PRESHUFFLED)
capi.run_timed(probs[(0, 0)], None, sp, 0, 1000)
t = {k: [] for k in probs}
for _ in range(7):
    for k, p in probs.items():
        t[k].append(capi.run_timed(p, None, sp, 0, 20))
for k, v in t.items():
    med = statistics.median(v)
    print(f"lda = K + {k[0]:3d}, ldc = N + {k[1]:2d}: {med * 1e3:7.1f} us  {2.0 * M * N * K / med / 1e9:7.1f} TFLOP/s")
