#!/usr/bin/env python3
"""A/B of two fp8 x int4 kernel variants selected by wqgemm_problem.flags: outputs compared (bit-equal on
integer-valued inputs, close on random bytes), then interleaved timing.

    python tools/f8_ab.py [flagsA] [flagsB] [M N K]
"""
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from composable_kernel_amd import capi  # noqa: E402

FA = int(sys.argv[1]) if len(sys.argv) > 1 else 0
FB = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda", 0)
sp = torch.cuda.current_stream().cuda_stream


def make(M, N, K, integer, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if integer:
        codes = torch.tensor([0x00, 0x38, 0x40, 0xB8, 0xC0, 0x44], dtype=torch.uint8, device=dev)  # 0 1 2 -1 -2 3
        a = codes[torch.randint(0, 6, (M, K), device=dev, generator=g)]
    else:
        a = torch.randint(0, 0x50, (M, K), device=dev, generator=g, dtype=torch.int16).to(torch.uint8)
        a |= (torch.randint(0, 2, (M, K), device=dev, generator=g, dtype=torch.int16).to(torch.uint8) << 7)
    q = torch.randint(-8, 8, (N, K), device=dev, generator=g, dtype=torch.int8)
    b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
    capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
    d0 = torch.rand((N,), device=dev, generator=g) * (1.0 if not integer else 0) + (1.0 if integer else 0.01)
    d1 = torch.rand((M,), device=dev, generator=g) * (1.0 if not integer else 0) + (1.0 if integer else 0.01)
    return a, b, d0, d1


def run(flags, a, b, d0, d1, M, N, K, epi):
    c = torch.zeros((M + 8, N), dtype=torch.float16, device=dev)
    p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_E4M3FN,
                             c_dtype=capi.F16, epilogue=epi, d0=d0.data_ptr(), d1=d1.data_ptr(), flags=flags)
    assert capi.is_supported(p), capi.lib.wqgemm_last_error()
    capi.launch(p, None, sp)
    torch.cuda.synchronize()
    assert (c[M:] == 0).all()
    return c[:M], p


shapes = [(256, 256, 128), (256, 256, 256), (256, 256, 384), (512, 256, 512), (300, 512, 640), (1024, 1024, 1024),
          (4096, 4096, 4096)]
if len(sys.argv) > 5:
    shapes = [tuple(int(x) for x in sys.argv[3:6])]
for (M, N, K) in shapes:
    for integer in (True, False):
        a, b, d0, d1 = make(M, N, K, integer, M + K)
        ca, _ = run(FA, a, b, d0, d1, M, N, K, capi.EPI_MULTIPLY_MULTIPLY)
        cb, _ = run(FB, a, b, d0, d1, M, N, K, capi.EPI_MULTIPLY_MULTIPLY)
        if integer:
            same = torch.equal(ca, cb)
            print(f"{M}x{N}x{K} integer: bit-equal={same}")
            assert same
        else:
            fin = torch.isfinite(ca.float()) & torch.isfinite(cb.float())
            err = (ca.float() - cb.float()).abs()[fin]
            tol = 1e-3 + 1e-3 * ca.float().abs()[fin]
            print(f"{M}x{N}x{K} random: max err {err.max().item():.4g}, outside tol {(err > tol).sum().item()}, "
                  f"bit-equal={torch.equal(ca.view(torch.int16), cb.view(torch.int16))}")
            assert (err <= 4 * tol).all()

M = N = K = 4096
a, b, d0, d1 = make(M, N, K, False, 1)
res = {}
probs = {f: run(f, a, b, d0, d1, M, N, K, capi.EPI_MULTIPLY_MULTIPLY_FASTGELU)[1] for f in (FA, FB)}
for f, p in probs.items():
    capi.run_timed(p, None, sp, 0, 2000)
for rep in range(5):
    for f, p in probs.items():
        res.setdefault(f, []).append(capi.run_timed(p, None, sp, 5, 200))
for f, ts in res.items():
    ms = statistics.median(ts)
    print(f"flags {f}: {ms * 1e3:.1f} us  {2 * M * N * K / ms / 1e9:.0f} TFLOP/s  ({capi.kernel_name(probs[f])})")
