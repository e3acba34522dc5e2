#!/usr/bin/env python3
"""A short fixed run of ONE problem, meant to sit under `rocprofv3 --pmc ... -- python3 tools/pmc_probe.py ...`.

    python3 tools/pmc_probe.py --a f16 --b i4 --M 1 --N 28672 --K 8192 [--launches 20]
    python3 tools/pmc_probe.py --a i8 --M 4096 --N 4096 --K 4096

Prints the kernel the planner chose and the algorithmic bytes per launch; the counters come from the profiler.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="f16", choices=["f16", "bf16", "i8"])
    ap.add_argument("--b", default="i4", choices=["i4", "i4pre", "i8", "i8pre"], help="i4pre / i8pre: pre-shuffled weight order")
    ap.add_argument("--M", type=int, default=1)
    ap.add_argument("--N", type=int, default=28672)
    ap.add_argument("--K", type=int, default=8192)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--flags", type=int, default=0, help="wqgemm_problem.flags (kernel variants)")
    args = ap.parse_args()
    M, N, K = args.M, args.N, args.K
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    sp = torch.cuda.current_stream().cuda_stream
    if args.a == "i8":
        a = torch.randint(-128, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
        b = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
        c = torch.empty((M, N), dtype=torch.int8, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.I8,
                                 b_dtype=capi.I8, c_dtype=capi.I8, epilogue=capi.EPI_ACT_MUL_CLAMP, requant=0.03)
        nbytes = M * K + N * K + M * N
    else:
        adt = torch.float16 if args.a == "f16" else torch.bfloat16
        a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).to(adt)
        s = (torch.rand((N,), generator=g, device=dev) * 2.0 ** -5 + 2.0 ** -8).to(adt)
        if args.b in ("i4", "i4pre"):
            q = torch.randint(-8, 8, (N, K), generator=g, device=dev, dtype=torch.int8)
            b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
            capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
            bdt, bb = capi.I4_PACKED, N * K // 2
            if args.b == "i4pre":
                b2 = torch.empty_like(b)
                capi.preshuffle_int4(b.data_ptr(), K // 2, K, N, b2.data_ptr(), K, sp)
                b, bdt = b2, capi.I4_PRESHUFFLED
        else:
            b = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
            bdt, bb = capi.I8, N * K
            if args.b == "i8pre":
                b2 = torch.empty_like(b)
                capi.preshuffle_int8(b.data_ptr(), K, K, N, b2.data_ptr(), K, sp)
                b, bdt = b2, capi.I8_PRESHUFFLED
        c = torch.empty((M, N), dtype=adt, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K,
                                 a_dtype=capi.F16 if args.a == "f16" else capi.BF16, b_dtype=bdt, flags=args.flags)
        nbytes = 2 * M * K + bb + 2 * N + 2 * M * N
    assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
    wsb = capi.workspace_bytes(p)
    ws_t = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for _ in range(args.launches):
        capi.launch(p, ws_t.data_ptr() if wsb else None, sp)
    torch.cuda.synchronize()
    print(f"kernel {capi.kernel_name(p)}  {capi.type_string(p)}  algorithmic_bytes_per_launch {nbytes}  "
          f"workspace {wsb}  launches {args.launches}")


if __name__ == "__main__":
    main()
