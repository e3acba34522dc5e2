#!/usr/bin/env python3
"""Diagnostic: per-section work / barrier-wait cycles of the default 256-tile kernel (workgroup 0)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402

M = N = 4096
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
F8 = len(sys.argv) > 3 and sys.argv[3] == "f8"
ZERO = len(sys.argv) > 4 and sys.argv[4] == "zero"  # all-zero operands: the same instruction stream at minimal switching power
dev = torch.device("cuda", 0)
a = (torch.rand((M, K), device=dev) * 2 - 1).half()
q = torch.randint(-8, 8, (N, K), device=dev, dtype=torch.int8)
s = (torch.rand((N,), device=dev) * 0.05 + 0.004).half()
if ZERO:
    a.zero_()
    q.zero_()
b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
sp = torch.cuda.current_stream().cuda_stream
capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
c = torch.empty((M, N), dtype=torch.float16, device=dev)
if F8:
    a = torch.randint(40, 80, (M, K), device=dev, dtype=torch.int16).to(torch.uint8)
    d0 = torch.rand((N,), device=dev)
    d1 = torch.rand((M,), device=dev)
    p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_E4M3FN,
                             c_dtype=capi.F16, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0.data_ptr(), d1=d1.data_ptr(),
                             flags=FLAGS)
else:
    p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, flags=FLAGS)
for _ in range(300):
    capi.launch(p, None, sp)
buf = torch.zeros(192, dtype=torch.int64, device=dev)
fn = capi.lib.wqgemm_debug_section_cycles
fn.restype = C.c_int
fn.argtypes = [C.POINTER(capi.Problem), C.c_void_p, C.c_void_p]
assert fn(C.byref(p), buf.data_ptr(), sp) == 0
torch.cuda.synchronize()
full = buf.cpu().numpy()
v = full[:64].reshape(8, 8)
nt = K // (128 if F8 else 64)
print("per K tile, cycles: wave | work L0 M0 L1 M1 | wait L0 M0 L1 M1 | total")
for w in range(8):
    r = v[w] / nt
    print(w, " ".join(f"{x:7.0f}" for x in r[:4]), "|", " ".join(f"{x:7.0f}" for x in r[4:]), "|", f"{r.sum():8.0f}")
if F8:
    print("L0 sub-phases per K tile (cycles from section start): wave | DMA issued | reads issued")
    for w in range(8):
        print(w, " ".join(f"{x / nt:7.0f}" for x in full[w * 8 + 64:w * 8 + 66]))
