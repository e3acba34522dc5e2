#!/usr/bin/env python3
"""int8 x int8 requantised GEMM: bit-equality and interleaved timing of kernel variants in one process.

    python tools/i8_ab.py [--M 4096 --N 4096 --K 4096] [--variants 0,8192,1] [--rounds 7] [--iters 20]

A variant is a value of wqgemm_problem.flags (0 = default route, 8192 = ping-pong form of the 256-tile
kernel, 1 = 128-tile kernel).  Every variant's E must equal the first variant's, byte for byte.
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--K", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ramp", type=int, default=1000)
    ap.add_argument("--variants", default="0,8192,1")
    ap.add_argument("--bias", type=int, default=1)
    args = ap.parse_args()
    M, N, K = args.M, args.N, args.K
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    a = torch.randint(-128, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
    b = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
    bias = torch.randint(-300000, 300001, (N,), generator=g, device=dev, dtype=torch.int32)
    sp = torch.cuda.current_stream().cuda_stream
    variants = [int(v) for v in args.variants.split(",")]
    outs, probs = {}, {}
    for v in variants:
        e = torch.full((M, N), 77, dtype=torch.int8, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, e.data_ptr(), M, N, K, a_dtype=capi.I8,
                                 b_dtype=capi.I8, c_dtype=capi.I8,
                                 epilogue=capi.EPI_ADD_RELU_MUL_CLAMP if args.bias else capi.EPI_ACT_MUL_CLAMP,
                                 requant=5e-5, d0=bias.data_ptr() if args.bias else None, flags=v)
        if not capi.is_supported(p):
            print(f"variant {v}: unsupported ({capi.lib.wqgemm_last_error().decode()})")
            continue
        capi.launch(p, None, sp)
        torch.cuda.synchronize()
        outs[v], probs[v] = e, p
    first = next(iter(outs))
    for v, e in outs.items():
        same = bool(torch.equal(e, outs[first]))
        print(f"flags={v:5d} {capi.kernel_name(probs[v]):22s} equal to flags={first}: {same}   "
              f"distinct values {int(torch.unique(e).numel())}")
    times = {v: [] for v in probs}
    for p in probs.values():
        capi.run_timed(p, None, sp, 3, 3)
    capi.run_timed(probs[first], None, sp, 0, max(1, args.ramp))
    for _ in range(args.rounds):
        for v, p in probs.items():
            times[v].append(capi.run_timed(p, None, sp, 0, args.iters))
    ops = 2.0 * M * N * K
    for v, ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        print(f"flags={v:5d} median {med * 1e3:8.1f} us {ops / med / 1e9:7.1f} TOP/s   min {mn * 1e3:8.1f} us "
              f"{ops / mn / 1e9:7.1f} TOP/s")


if __name__ == "__main__":
    main()
