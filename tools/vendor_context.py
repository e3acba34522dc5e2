#!/usr/bin/env python3
"""Context numbers: the vendor libraries' plain GEMMs (no dequantisation, operands already in the matrix core's
type) at the benchmark shapes, timed on the same GPU with the same method as tools/config_bench.py (clock settling,
median of 7 x 30 launches between events).  They are not baselines of this path -- the reference publishes none for
MI355X -- but say what a tuned dense GEMM of the same arithmetic type reaches on the box the numbers were taken on.

    python tools/vendor_context.py [--out profiles/r1_vendor_context.jsonl]
"""
import argparse
import json
import statistics

import torch


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run(n):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    one = run(10)
    run(max(50, int(0.25 / max(one * 1e-3, 1e-6))))
    return statistics.median(run(30) for _ in range(7))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lines = []

    def emit(name, M, N, K, fn, peak):
        try:
            ms = timed(fn)
        except Exception as exc:  # a library without this GEMM on this build: record that, carry on
            line = {"gemm": name, "M": M, "N": N, "K": K, "error": f"{type(exc).__name__}: {exc}"[:200]}
        else:
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            line = {"gemm": name, "M": M, "N": N, "K": K, "ms": ms, "tflops": tf, "mfma_peak_tflops": peak,
                    "frac_of_mfma_peak": tf / peak}
        print(json.dumps(line), flush=True)
        lines.append(line)

    for dt, tag in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
        for (M, N, K) in ((4096, 4096, 4096), (4096, 11008, 4096), (4096, 4096, 8192), (8192, 8192, 8192)):
            a = torch.randn((M, K), device=dev, dtype=dt)
            b = torch.randn((N, K), device=dev, dtype=dt)
            c = torch.empty((M, N), device=dev, dtype=dt)
            emit(f"torch.matmul {tag} (A row-major, B^T: K contiguous)", M, N, K,
                 lambda a=a, b=b, c=c: torch.matmul(a, b.t(), out=c), 2500.0)
    # the same GEMMs on the benchmark's own value distribution (A ~ U(-1,1), B = int4 weight * per-channel scale,
    # already dequantised): matrix-core power, and with it the clock, depends on the operand bits
    g = torch.Generator(device=dev)
    g.manual_seed(11939)
    for dt, tag in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
        for (M, N, K) in ((4096, 4096, 4096), (4096, 11008, 4096)):
            a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).to(dt)
            sc = torch.rand((N, 1), generator=g, device=dev) * (2.0 ** -4 - 2.0 ** -8) + 2.0 ** -8
            b = (torch.randint(-8, 8, (N, K), generator=g, device=dev).float() * sc).to(dt)
            c = torch.empty((M, N), device=dev, dtype=dt)
            emit(f"torch.matmul {tag}, benchmark value distribution (dequantised int4 weights)", M, N, K,
                 lambda a=a, b=b, c=c: torch.matmul(a, b.t(), out=c), 2500.0)
    M = N = K = 4096
    a8 = torch.randint(-128, 128, (M, K), device=dev, dtype=torch.int8)
    b8 = torch.randint(-128, 128, (N, K), device=dev, dtype=torch.int8)
    emit("torch._int_mm int8 -> int32", M, N, K, lambda: torch._int_mm(a8, b8.t()), 5000.0)
    for f8, tag in ((torch.float8_e4m3fn, "e4m3fn"), (torch.float8_e4m3fnuz, "e4m3fnuz")):
        try:
            af = torch.randn((M, K), device=dev).to(f8)
            bf = torch.randn((N, K), device=dev).to(f8)
            one = torch.ones((), device=dev)
            emit(f"torch._scaled_mm fp8 {tag} -> fp16", M, N, K,
                 lambda af=af, bf=bf: torch._scaled_mm(af, bf.t(), scale_a=one, scale_b=one, out_dtype=torch.float16),
                 5000.0)
        except Exception as exc:
            print(json.dumps({"gemm": f"fp8 {tag}", "error": str(exc)[:200]}), flush=True)
    if args.out:
        with open(args.out, "w") as fh:
            for line in lines:
                fh.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
