#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (variants x rounds), on random data.

    python tools/ab_bench.py [--M 4096 --N 4096 --K 4096] [--rounds 7] [--iters 20] [--variants 0,4,1]

A variant is a value of wqgemm_problem.flags (0 = default route, 1 = 128-tile, 2 = 256-tile,
4 = 256-tile lockstep schedule, combinations allowed).  Prints median / min ms and TFLOP/s.
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
    ap.add_argument("--variants", default="0,4,1")
    ap.add_argument("--a", default="f16", choices=["f16", "bf16", "f8", "f8fnuz"])
    ap.add_argument("--ramp", type=int, default=1000, help="untimed launches before timing (clock settling)")
    ap.add_argument("--b", default="i4", choices=["i4", "i4pre", "i8", "i8pre", "u8"], help="i4pre / i8pre: the weights in their pre-shuffled order")
    ap.add_argument("--group-k", type=int, default=0, help="group-wise weight scales: k per group (fp16 activations)")
    args = ap.parse_args()
    M, N, K = args.M, args.N, args.K
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    is_f8 = args.a in ("f8", "f8fnuz")
    adt = torch.float16 if args.a in ("f16", "f8", "f8fnuz") else torch.bfloat16
    if is_f8:
        # random fp8 bytes with exponent fields 1..14 (no NaN / maximum encodings in either E4M3 flavour)
        a = (torch.randint(8, 120, (M, K), generator=g, device=dev, dtype=torch.int16) +
             128 * torch.randint(0, 2, (M, K), generator=g, device=dev, dtype=torch.int16)).to(torch.uint8)
        d0 = torch.rand((N,), generator=g, device=dev)
        d1 = torch.rand((M,), generator=g, device=dev)
    else:
        a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).to(adt)
    s_rows = K // args.group_k if args.group_k else 1
    s = (torch.rand((s_rows, N), generator=g, device=dev) * (2.0 ** -4 - 2.0 ** -8) + 2.0 ** -8).to(adt)
    if args.b in ("i4", "i4pre"):
        q = torch.randint(-8, 8, (N, K), generator=g, device=dev, dtype=torch.int8)
        b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
        capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, torch.cuda.current_stream().cuda_stream)
        bdt = capi.I4_PACKED
        if args.b == "i4pre":
            b2 = torch.empty_like(b)
            capi.preshuffle_int4(b.data_ptr(), K // 2, K, N, b2.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
            b, bdt = b2, capi.I4_PRESHUFFLED
    else:
        b = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
        bdt = capi.I8 if args.b in ("i8", "i8pre") else capi.U8_OFFSET
        if args.b == "i8pre":
            b2 = torch.empty_like(b)
            capi.preshuffle_int8(b.data_ptr(), K, K, N, b2.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
            b, bdt = b2, capi.I8_PRESHUFFLED
    c = torch.empty((M, N), dtype=adt, device=dev)
    torch.cuda.synchronize()
    sp = torch.cuda.current_stream().cuda_stream
    variants = [int(v) for v in args.variants.split(",")]
    probs = {}
    for v in variants:
        if is_f8:
            p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K,
                                     a_dtype=capi.F8_E4M3FN if args.a == "f8" else capi.F8_FNUZ, b_dtype=bdt,
                                     c_dtype=capi.F16, epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU,
                                     d0=d0.data_ptr(), d1=d1.data_ptr(), flags=v)
        else:
            p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K,
                                     a_dtype=capi.F16 if args.a == "f16" else capi.BF16, b_dtype=bdt, flags=v,
                                     scale_group_k=args.group_k)
        if not capi.is_supported(p):
            print(f"variant {v}: unsupported ({capi.lib.wqgemm_last_error().decode()})")
            continue
        probs[v] = p
    times = {v: [] for v in probs}
    ws_bytes = max([capi.workspace_bytes(p) for p in probs.values()] + [0])
    ws_t = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    ws = ws_t.data_ptr() if ws_t is not None else None
    for v, p in probs.items():          # warm-up
        capi.run_timed(p, ws, sp, 3, 3)
    if probs:
        capi.run_timed(next(iter(probs.values())), ws, sp, 0, max(1, args.ramp))
    for _ in range(args.rounds):
        for v, p in probs.items():
            times[v].append(capi.run_timed(p, ws, sp, 0, args.iters))
    flops = 2.0 * M * N * K
    for v, ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        print(f"flags={v:2d} {capi.kernel_name(probs[v]):18s} median {med * 1e3:8.1f} us  {flops / med / 1e9:7.1f} TF   "
              f"min {mn * 1e3:8.1f} us  {flops / mn / 1e9:7.1f} TF")


if __name__ == "__main__":
    main()
