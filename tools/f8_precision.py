#!/usr/bin/env python3
"""Diagnostic: raw fp32 accumulators of the fp8 MFMA kernel against an exact float64 sum, for byte sets of
different dynamic range."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402


def run(a, q, M, N, K, dtype):
    dev = torch.device("cuda", 0)
    a_t = torch.from_numpy(a).to(dev)
    q_t = torch.from_numpy(q).to(dev)
    b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
    sp = torch.cuda.current_stream().cuda_stream
    capi.pack_int4(q_t.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
    c = torch.zeros((M, N), dtype=torch.float32, device=dev)
    p = capi.fpAintB_problem(a_t.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=dtype,
                             b_dtype=capi.I4_PACKED, c_dtype=capi.F32, k_begin=0, k_end=K,
                             flags=capi.FLAG_F8_NO_SPECIALS)
    assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
    name = capi.kernel_name(p)
    capi.launch(p, None, sp)
    torch.cuda.synchronize()
    return c.cpu().numpy(), name


def main():
    rng = np.random.default_rng(0)
    M = N = 256
    for K in (128, 1024):
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        for label, lo, hi in (("exp 7..8", 56, 72), ("exp 4..11", 32, 96), ("exp 1..14", 8, 120), ("all finite", 0, 127)):
            mag = rng.integers(lo, hi, (M, K))
            a = (mag + 128 * rng.integers(0, 2, (M, K))).astype(np.uint8)
            a[a == 0x80] = 0
            got, name = run(a, q, M, N, K, capi.F8_E4M3FN)
            a_f = torch.from_numpy(a).view(torch.float8_e4m3fn).to(torch.float64).numpy()
            ref = a_f @ q.astype(np.float64).T
            absmax = (np.abs(a_f) @ np.abs(q.astype(np.float64)).T)
            err = np.abs(got - ref)
            print(f"K={K:5d} {label:10s} {name}: max|err|/sum|terms| = {(err / absmax).max():.3e}   "
                  f"max|err|/|ref| (|ref|>1% of terms) = {(err / np.abs(ref))[np.abs(ref) > 0.01 * absmax].max():.3e}")


if __name__ == "__main__":
    main()
