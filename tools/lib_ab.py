#!/usr/bin/env python3
"""Interleaved timing of TWO BUILDS of libwqgemm.so in one process, same operands, same problem:

    python tools/lib_ab.py --base tools/_ab/libwqgemm_base.so [--new composable_kernel_amd/csrc/libwqgemm.so]
                           [--configs C2,C3,C5,C5fn,I8] [--rounds 9] [--iters 30] [--K 4096]

Both libraries are loaded side by side (their kernels live in separate code objects); each round times `iters`
launches of one build, then of the other.  Prints the median time per build, the ratio, and whether the two outputs
are bit-identical.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from composable_kernel_amd import capi  # noqa: E402


def bind(path):
    lib = C.CDLL(os.path.abspath(path))
    P = C.POINTER(capi.Problem)
    lib.wqgemm_launch.restype = C.c_int
    lib.wqgemm_launch.argtypes = [P, C.c_void_p, C.c_void_p]
    lib.wqgemm_run_timed.restype = C.c_int
    lib.wqgemm_run_timed.argtypes = [P, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.wqgemm_last_error.restype = C.c_char_p
    lib.wqgemm_workspace_bytes.restype = C.c_size_t
    lib.wqgemm_workspace_bytes.argtypes = [P]
    return lib


def timed(lib, p, ws, sp, iters):
    ms = C.c_float(0)
    rc = lib.wqgemm_run_timed(C.byref(p), ws, sp, 0, iters, C.byref(ms))
    assert rc == 0, lib.wqgemm_last_error()
    return ms.value * 1e3


def make(cfg, K, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    sp = torch.cuda.current_stream().cuda_stream
    M = N = 4096
    if cfg == "C3":
        N = 11008
        a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).bfloat16()
        s = (torch.rand((N,), generator=g, device=dev) * (2.0 ** -6 - 2.0 ** -10) + 2.0 ** -10).bfloat16()
        b = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
        bpre = torch.empty_like(b)
        capi.preshuffle_int8(b.data_ptr(), K, K, N, bpre.data_ptr(), K, sp)
        c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), bpre.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, a_dtype=capi.BF16,
                                 b_dtype=capi.I8_PRESHUFFLED)
        return p, c, (a, s, b, bpre), 2.0 * M * N * K
    if cfg == "I8":
        a = torch.randint(-128, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
        b = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
        c = torch.empty((M, N), dtype=torch.int8, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.I8, b_dtype=capi.I8,
                                 c_dtype=capi.I8, epilogue=capi.EPI_ACT_MUL_CLAMP, requant=0.0003)
        return p, c, (a, b), 2.0 * M * N * K
    q = torch.randint(-8, 8, (N, K), generator=g, device=dev, dtype=torch.int8)
    bp = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
    capi.pack_int4(q.data_ptr(), K, K, N, bp.data_ptr(), K // 2, sp)
    bpre = torch.empty_like(bp)
    capi.preshuffle_int4(bp.data_ptr(), K // 2, K, N, bpre.data_ptr(), K, sp)
    if cfg == "C2":
        a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).half()
        s = (torch.rand((N,), generator=g, device=dev) * (2.0 ** -4 - 2.0 ** -8) + 2.0 ** -8).half()
        c = torch.empty((M, N), dtype=torch.float16, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), bpre.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)
        return p, c, (a, s, q, bp, bpre), 2.0 * M * N * K
    # C5 (FNUZ bytes, no specials) / C5fn (E4M3FN bytes)
    a = (torch.randint(24, 88, (M, K), generator=g, device=dev, dtype=torch.int16) +
         128 * torch.randint(0, 2, (M, K), generator=g, device=dev, dtype=torch.int16)).to(torch.uint8)
    d0 = torch.rand((N,), generator=g, device=dev) * 2.0 ** -4
    d1 = torch.rand((M,), generator=g, device=dev)
    c = torch.empty((M, N), dtype=torch.float16, device=dev)
    fn = cfg == "C5fn"
    p = capi.fpAintB_problem(a.data_ptr(), bpre.data_ptr(), None, c.data_ptr(), M, N, K,
                             a_dtype=capi.F8_E4M3FN if fn else capi.F8_FNUZ, b_dtype=capi.I4_PRESHUFFLED, c_dtype=capi.F16,
                             epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU, d0=d0.data_ptr(), d1=d1.data_ptr(),
                             flags=0 if fn else capi.FLAG_F8_NO_SPECIALS)
    return p, c, (a, q, bp, bpre, d0, d1), 2.0 * M * N * K


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", required=True)
    ap.add_argument("--new", default=os.path.join(ROOT, "composable_kernel_amd", "csrc", "libwqgemm.so"))
    ap.add_argument("--configs", default="C2,C3,C5,C5fn,I8")
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--K", type=int, default=4096)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    libs = {"base": bind(args.base), "new": bind(args.new)}
    for cfg in args.configs.split(","):
        p, c, keep, flops = make(cfg, args.K, dev)
        outs = {}
        for name, lib in libs.items():
            c.zero_()
            rc = lib.wqgemm_launch(C.byref(p), None, sp)
            assert rc == 0, lib.wqgemm_last_error()
            torch.cuda.synchronize()
            outs[name] = c.clone()
        same = bool(torch.equal(outs["base"].view(torch.uint8), outs["new"].view(torch.uint8)))
        for lib in libs.values():   # clocks
            timed(lib, p, None, sp, 400)
        t = {k: [] for k in libs}
        for _ in range(args.rounds):
            for name, lib in libs.items():
                t[name].append(timed(lib, p, None, sp, args.iters))
        mb, mn = statistics.median(t["base"]), statistics.median(t["new"])
        print(f"{cfg:5s} K={args.K}: base {mb:8.2f} us ({flops / mb / 1e6:7.1f} T)   new {mn:8.2f} us ({flops / mn / 1e6:7.1f} T)   "
              f"new/base time {mn / mb:.4f}   same bits: {same}", flush=True)
        del keep, outs


if __name__ == "__main__":
    main()
