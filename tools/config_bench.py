#!/usr/bin/env python3
"""One JSON line per BASELINE configuration (1 GPU): kernel, time, TFLOP/s on 2*M*N*K, algorithmic GB/s and
the fraction of the MFMA roofline of the arithmetic type.  Inputs are synthetic and resident in HBM.

    python tools/config_bench.py [--out profiles/r1_configs.jsonl]

Timing: ~0.2 s of untimed launches (clock settling), then the median of 7 x 30 launches between HIP events
(wqgemm_run_timed).  The per-GPU slice of the split-K configuration is timed as one K-slice launch that
stores the fp32 partial.
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402

PEAK = {"f16": 2500.0, "bf16": 2500.0, "f8": 5000.0, "i8": 5000.0}  # dense TFLOP/s (TOP/s), vendor nominal
PEAK_HBM = 8000.0


def timed(p, ws, sp, flops):
    capi.run_timed(p, ws, sp, 3, 3)
    capi.run_timed(p, ws, sp, 0, max(50, int(0.25 / max(capi.run_timed(p, ws, sp, 0, 10) * 1e-3, 1e-6))))
    ts = [capi.run_timed(p, ws, sp, 0, 30) for _ in range(7)]
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11939)
    sp = torch.cuda.current_stream().cuda_stream
    lines = []

    def emit(name, arith, p, ws_t, M, N, K, nbytes, note=""):
        ws = ws_t.data_ptr() if ws_t is not None else None
        ms = timed(p, ws, sp, 2.0 * M * N * K)
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        line = {"config": name, "M": M, "N": N, "K": K, "kernel": capi.kernel_name(p), "instance": capi.type_string(p),
                "ms": ms, "tflops": tf, "mfma_peak_tflops": PEAK[arith], "frac_of_mfma_peak": tf / PEAK[arith],
                "algorithmic_bytes": nbytes, "hbm_gbps_algorithmic": nbytes / (ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM, "data": "synthetic", "note": note}
        print(json.dumps(line), flush=True)
        lines.append(line)

    def packed(N, K):
        q = torch.randint(-8, 8, (N, K), generator=g, device=dev, dtype=torch.int8)
        b = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
        capi.pack_int4(q.data_ptr(), K, K, N, b.data_ptr(), K // 2, sp)
        return b

    # C1: int8 x int8 requant 256^3 (example/14's verification size) and 1024^3 / 4096^3
    for S in (256, 1024, 4096):
        a = torch.randint(-128, 128, (S, S), generator=g, device=dev, dtype=torch.int8)
        b = torch.randint(-128, 128, (S, S), generator=g, device=dev, dtype=torch.int8)
        e = torch.empty((S, S), dtype=torch.int8, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, e.data_ptr(), S, S, S, a_dtype=capi.I8, b_dtype=capi.I8,
                                 c_dtype=capi.I8, epilogue=capi.EPI_ACT_MUL_CLAMP, requant=0.03)
        emit(f"C1 int8 x int8 requant {S}^3", "i8", p, None, S, S, S, 3 * S * S)

    # C2: fp16 x int4 4096^3
    M = N = K = 4096
    a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).half()
    s = (torch.rand((N,), generator=g, device=dev) * (2.0 ** -4 - 2.0 ** -8) + 2.0 ** -8).half()
    b = packed(N, K)
    c = torch.empty((M, N), dtype=torch.float16, device=dev)
    p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K)
    emit("C2 fp16 x int4 4096^3", "f16", p, None, M, N, K, 2 * M * K + K * N // 2 + 2 * N + 2 * M * N)

    # C3: bf16 x int8 4096 x 11008 x 4096, PassThrough and FastGelu
    M, N, K = 4096, 11008, 4096
    a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).bfloat16()
    s = (torch.rand((N,), generator=g, device=dev) * (2.0 ** -6 - 2.0 ** -10) + 2.0 ** -10).bfloat16()
    b8 = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
    c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for epi, tag in ((capi.EPI_PASSTHROUGH, "PassThrough"), (capi.EPI_FASTGELU, "FastGelu")):
        p = capi.fpAintB_problem(a.data_ptr(), b8.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, a_dtype=capi.BF16,
                                 b_dtype=capi.I8, epilogue=epi)
        emit(f"C3 bf16 x int8 4096x11008x4096 {tag}", "bf16", p, None, M, N, K, 2 * M * K + K * N + 2 * N + 2 * M * N)

    # C4: per-GPU slice of the 8-GPU split-K (K = 65536 / 8), fp32 partial out
    M = N = 4096
    K = 8192
    a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).half()
    s = (torch.rand((N,), generator=g, device=dev) * (2.0 ** -4 - 2.0 ** -8) + 2.0 ** -8).half()
    b = packed(N, K)
    part = torch.empty((M, N), dtype=torch.float32, device=dev)
    p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), s.data_ptr(), part.data_ptr(), M, N, K, k_begin=0, k_end=K,
                             c_dtype=capi.F32)
    emit("C4 fp16 x int4 split-K slice 4096x4096x8192 (1 of 8 GPUs)", "f16", p, None, M, N, K,
         2 * M * K + K * N // 2 + 2 * N + 4 * M * N)

    # C5: fp8 x int4 + MultiplyMultiply + FastGelu
    for (M, N, K) in ((4096, 4096, 4096), (3840, 4096, 4096)):
        a = (torch.randint(24, 88, (M, K), generator=g, device=dev, dtype=torch.int16) +
             128 * torch.randint(0, 2, (M, K), generator=g, device=dev, dtype=torch.int16)).to(torch.uint8)
        b = packed(N, K)
        d0 = torch.rand((N,), generator=g, device=dev)
        d1 = torch.rand((M,), generator=g, device=dev)
        c = torch.empty((M, N), dtype=torch.float16, device=dev)
        nbytes = M * K + K * N // 2 + 4 * (M + N) + 2 * M * N
        for dt, flags, tag, arith in ((capi.F8_E4M3FN, 0, "E4M3FN", "f8"),
                                      (capi.F8_FNUZ, capi.FLAG_F8_NO_SPECIALS, "FNUZ, no special encodings", "f8"),
                                      (capi.F8_FNUZ, 0, "FNUZ, any encoding (fp16 image)", "f16")):
            p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=dt, c_dtype=capi.F16,
                                     epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU, d0=d0.data_ptr(), d1=d1.data_ptr(),
                                     flags=flags)
            nb = capi.workspace_bytes(p)
            ws_t = torch.empty(nb, dtype=torch.uint8, device=dev) if nb else None
            emit(f"C5 fp8 x int4 {M}x{N}x{K} MultiplyMultiply+FastGelu ({tag})", arith, p, ws_t, M, N, K, nbytes)

    # fp8 x int4 with a scale per 128 x 128 block of both operands (example/65's ..._fp8_ab_scale shape, bf16 out)
    for (M, N, K) in ((4096, 4096, 4096), (3840, 4096, 4096)):
        a = (torch.randint(24, 88, (M, K), generator=g, device=dev, dtype=torch.int16) +
             128 * torch.randint(0, 2, (M, K), generator=g, device=dev, dtype=torch.int16)).to(torch.uint8)
        b = packed(N, K)
        sa = torch.rand(((M + 127) // 128, K // 128), generator=g, device=dev)
        sb = torch.rand((N // 128, K // 128), generator=g, device=dev)
        c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        nbytes = M * K + K * N // 2 + 4 * (sa.numel() + sb.numel()) + 2 * M * N
        for dt, flags, tag in ((capi.F8_E4M3FN, 0, "E4M3FN"), (capi.F8_FNUZ, capi.FLAG_F8_NO_SPECIALS, "FNUZ, no special encodings")):
            p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=dt, c_dtype=capi.BF16,
                                     flags=flags, a_block_scale=sa.data_ptr(), b_block_scale=sb.data_ptr(),
                                     scale_block=(128, 128, 128))
            emit(f"fp8 x int4 {M}x{N}x{K}, 128x128x128 block scales of A and B ({tag})", "f8", p, None, M, N, K, nbytes)
        # the same with one activation scale per token and 128 k (scale_block_m = 1)
        sa1 = torch.rand((M, K // 128), generator=g, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_E4M3FN, c_dtype=capi.BF16,
                                 a_block_scale=sa1.data_ptr(), b_block_scale=sb.data_ptr(), scale_block=(1, 128, 128))
        emit(f"fp8 x int4 {M}x{N}x{K}, 1x128 activation scales, 128x128 weight scales (E4M3FN)", "f8", p, None, M, N, K,
             nbytes + 4 * (sa1.numel() - sa.numel()))

    if args.out:
        with open(args.out, "w") as fh:
            for line in lines:
                fh.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
