#!/usr/bin/env python3
"""Benchmark of the weight-quantized GEMM hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N ...            (starts its own N ranks, one process per GPU, and prints rank 0's line)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (ranks started for it)

N = 1  : BASELINE config 2 -- fp16 A x packed-int4 B, per-channel fp16 scale, M = N = K = 4096.
         One step = one GEMM launch over one synthetic batch resident in HBM.
         --config C1 | C3 | C5 runs one of the other single-GPU BASELINE configurations the same way
         (int8 requant 256^3; bf16 x int8 4096 x 11008 x 4096; fp8 x int4 + MultiplyMultiply + FastGelu 4096^3).
N > 1  : BASELINE config 4 -- the same GEMM at M = N = 4096, K = 65536, K split across the N
         ranks (one process per GPU), fp32 partials summed with RCCL (reduce-scatter over xGMI),
         convert/epilogue on each rank's rows, all-gather of the fp16 rows.  Total work is fixed
         as N grows ("strong" scaling).

Rank 0 prints ONE JSON line.  `value` is whole-job TFLOP/s on 2*M*N*K with inputs resident in HBM.
`roofline` prices the dominant kernel (the MFMA GEMM) against the dense fp16 MFMA peak using HIP
events on the launch stream; `cpu_baseline` times the CPU checker (the plain-C restatement of
CK's host ReferencefpAintBGemm) on a bounded row block of the same inputs and cross-checks the GPU
rows against it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from composable_kernel_amd import capi, splitk  # noqa: E402

PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak of MI355X
PEAK_HBM_GBPS = 8000.0


def synth_inputs(M: int, N: int, K: int, seed: int, device):
    """Synthetic matrices of the named shape: A ~ U(-1,1) fp16, q uniform over the int4 range,
    scale ~ U(2^-8, 2^-4) fp16 (the "realistic" input set)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    a = (torch.rand((M, K), generator=g, device=device, dtype=torch.float32) * 2 - 1).to(torch.float16)
    q = torch.randint(-8, 8, (N, K), generator=g, device=device, dtype=torch.int8)
    s = (torch.rand((N,), generator=g, device=device, dtype=torch.float32) * (2.0 ** -4 - 2.0 ** -8)
         + 2.0 ** -8).to(torch.float16)
    bp = torch.empty((N, K // 2), dtype=torch.uint8, device=device)
    capi.pack_int4(q.data_ptr(), K, K, N, bp.data_ptr(), K // 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return a, q, s, bp


def preshuffled(bp: torch.Tensor, K: int, N: int) -> torch.Tensor:
    """The packed int4 weights re-ordered offline into WQ_I4_PRESHUFFLED (same nibbles; every wave-wide weight fetch of
    the 256-tile kernels becomes one contiguous kilobyte).  Same kernels, same arithmetic, same output bits."""
    out = torch.empty_like(bp)
    capi.preshuffle_int4(bp.data_ptr(), K // 2, K, N, out.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


def algorithmic_bytes(M, N, K, c_bytes=2):
    # sz(A) M K + 0.5 K N + sz(scale) N + sz(C) M N
    return 2 * M * K + K * N // 2 + 2 * N + c_bytes * M * N


def cpu_baseline_leg(a, q, s, c_gpu, M, N, K, budget_s=12.0):
    """Time the CPU checker on a bounded row block of the same inputs and compare the GPU rows."""
    from oracle import binding  # the checker: used here only as the timed baseline / cross-check
    port = binding.load_port()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(16, cores))
    a_h = a.cpu().numpy().view(np.uint16)
    q_h = q.cpu().numpy()
    s_h = s.cpu().numpy().view(np.uint16)
    # calibrate on 4 rows, then size the block for ~budget_s (at most 1024 rows)
    t0 = time.perf_counter()
    port.fpAintB_gemm_f16(np.ascontiguousarray(a_h[:4]), q_h, s_h, 4, N, K, K, K, N, nthreads=cores)
    t_cal = time.perf_counter() - t0
    rows = int(max(8, min(1024, (budget_s / max(t_cal, 1e-6)) * 4)))
    rows -= rows % 8
    t0 = time.perf_counter()
    ref = port.fpAintB_gemm_f16(np.ascontiguousarray(a_h[:rows]), q_h, s_h, rows, N, K, K, K, N, nthreads=cores)
    dt = time.perf_counter() - t0
    ref = ref.reshape(rows, N).view(np.float16).astype(np.float32)
    got = c_gpu[:rows].cpu().numpy().view(np.float16).astype(np.float32)
    err = np.abs(got - ref)
    ok = bool((err <= 1e-3 + 1e-3 * np.abs(ref)).all())
    return {
        "value": 2.0 * rows * N * K / dt / 1e12,
        "unit": "TFLOP/s",
        "cores": cores,
        "kind": "port",
        "sample": f"rows 0..{rows - 1} of A x all N={N} x full K={K} ({rows}/{M} of the GEMM), "
                  f"{dt:.2f} s; full-size estimate {dt * M / rows:.0f} s",
        "gpu_rows_match_checker": ok,
        "max_abs_err": float(err.max()),
    }


def pmc_traffic(kernel, config="C2"):
    """HBM bytes per launch of `kernel` from the committed counter summary (separate rocprofv3 --pmc passes for
    FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled as gfx950 requires; tools/pmc_summary.py), or None when the
    summary is absent or belongs to another kernel.  Counters cannot be collected from inside this process."""
    for name in PMC_SUMMARIES.get(config, ()):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        try:
            with open(path) as fh:
                summary = json.load(fh)
        except OSError:
            continue
        if any(kernel in k for k in summary.get("kernel", [])) and summary.get("hbm_bytes_per_launch"):
            return summary["hbm_bytes_per_launch"], (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an "
                                                     "earlier run of this kernel on this shape, not measured by this process")
    return None, None


PEAK_8BIT_MFMA_TFLOPS = 5000.0  # dense fp8 / int8 MFMA peak of MI355X
# counter summaries per configuration (each taken on that configuration's shape and weight order), newest first
PMC_SUMMARIES = {"C2": ("r3_c2_pmc.json", "r2_mfma256_preshuffled_pmc.json", "r2_mfma256_pmc.json", "r1_mfma256_pmc.json"),
                 "C3": ("r3_c3_pmc.json",), "C5": ("r3_f8_pmc.json",)}


def _cpu_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(16, cores))


def _row_block(run_rows, M, budget_s, probe=4):
    """Time run_rows(r) (the CPU checker on the first r rows of A) on a block sized for ~budget_s of work."""
    t0 = time.perf_counter()
    run_rows(probe)
    t_cal = time.perf_counter() - t0
    rows = int(max(8, min(1024, M, (budget_s / max(t_cal, 1e-6)) * probe)))
    rows -= rows % 8
    t0 = time.perf_counter()
    ref = run_rows(rows)
    return rows, time.perf_counter() - t0, ref


def workload_c2(dev):
    """BASELINE config 2: fp16 A x packed-int4 B, per-channel fp16 scale, 4096^3 (the headline workload)."""
    M = N = K = 4096
    a, q, s, bp = synth_inputs(M, N, K, 11939, dev)
    bpre = preshuffled(bp, K, N)  # the weights as an offline packer stores them for this engine
    c = torch.empty((M, N), dtype=torch.float16, device=dev)
    p = capi.fpAintB_problem(a.data_ptr(), bpre.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)

    def clone():
        a2, b2, c2 = a.clone(), bpre.clone(), torch.empty_like(c)
        return (a2, b2, c2), capi.fpAintB_problem(a2.data_ptr(), b2.data_ptr(), s.data_ptr(), c2.data_ptr(), M, N, K,
                                                  b_dtype=capi.I4_PRESHUFFLED)

    def cpu(budget_s):
        return cpu_baseline_leg(a, q, s, c, M, N, K, budget_s)

    return {"M": M, "N": N, "K": K, "problem": p, "keep": (a, q, s, bp, c, bpre), "clone": clone, "cpu": cpu,
            "dtype": "f16", "peak": PEAK_FP16_MFMA_TFLOPS, "bytes": algorithmic_bytes(M, N, K), "workspace": None,
            "workload": "fp16A x int4B GEMM M=N=K=4096, per-channel fp16 scale, fp16 C "
                        "(example/64_fpAintB_gemm shape class, BASELINE config 2); int4 weights packed two per byte "
                        "and stored offline in the engine's pre-shuffled order"}


def workload_c1(dev):
    """BASELINE config 1: int8 x int8 256^3 with Activation_Mul_Clamp(0.03) -- example/14's verification case, the
    reference's own CPU-runnable configuration.  Inputs uniform over the int8 range; the CPU leg runs the full size."""
    M = N = K = 256
    g = torch.Generator(device=dev)
    g.manual_seed(11939)
    a = torch.randint(-128, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
    b = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
    e = torch.empty((M, N), dtype=torch.int8, device=dev)

    def prob(a_, b_, e_):
        return capi.fpAintB_problem(a_.data_ptr(), b_.data_ptr(), None, e_.data_ptr(), M, N, K, a_dtype=capi.I8,
                                    b_dtype=capi.I8, c_dtype=capi.I8, epilogue=capi.EPI_ACT_MUL_CLAMP, requant=0.03)

    def clone():
        t = (a.clone(), b.clone(), torch.empty_like(e))
        return t, prob(*t)

    def cpu(budget_s):
        from oracle import binding  # the checker: timed baseline and cross-check only
        port = binding.load_port()
        cores = _cpu_cores()
        a_h, b_h = a.cpu().numpy(), b.cpu().numpy()
        port.gemm_i8_requant(a_h, b_h, M, N, K, K, K, N, 0.03, nthreads=cores)
        reps, t0 = 0, time.perf_counter()
        while True:
            ref = port.gemm_i8_requant(a_h, b_h, M, N, K, K, K, N, 0.03, nthreads=cores)
            reps += 1
            if time.perf_counter() - t0 > min(budget_s, 3.0):
                break
        dt = (time.perf_counter() - t0) / reps
        ok = bool((e.cpu().numpy().ravel() == ref).all())
        return {"value": 2.0 * M * N * K / dt / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
                "sample": f"the full 256^3 problem, mean of {reps} runs of {dt * 1e3:.2f} ms",
                "gpu_rows_match_checker": ok, "max_abs_err": 0.0 if ok else 1.0}

    return {"M": M, "N": N, "K": K, "problem": prob(a, b, e), "keep": (a, b, e), "clone": clone, "cpu": cpu,
            "dtype": "i8", "peak": PEAK_8BIT_MFMA_TFLOPS, "bytes": 3 * M * N, "workspace": None,
            "workload": "int8 x int8 GEMM M=N=K=256 + Activation_Mul_Clamp(0.03), int8 out "
                        "(example/14_gemm_quantization verification size, BASELINE config 1)"}


def workload_c3(dev):
    """BASELINE config 3: bf16 A x int8 B, per-channel bf16 scale, M=4096 N=11008 K=4096 (example/60's arithmetic)."""
    M, N, K = 4096, 11008, 4096
    g = torch.Generator(device=dev)
    g.manual_seed(11939)
    a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).bfloat16()
    s = (torch.rand((N,), generator=g, device=dev) * (2.0 ** -6 - 2.0 ** -10) + 2.0 ** -10).bfloat16()
    b = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
    bpre = torch.empty_like(b)  # the same bytes in the engine's pre-shuffled order (offline step; same output bits)
    capi.preshuffle_int8(b.data_ptr(), K, K, N, bpre.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

    def prob(a_, b_, c_):
        return capi.fpAintB_problem(a_.data_ptr(), b_.data_ptr(), s.data_ptr(), c_.data_ptr(), M, N, K,
                                    a_dtype=capi.BF16, b_dtype=capi.I8_PRESHUFFLED)

    def clone():
        t = (a.clone(), bpre.clone(), torch.empty_like(c))
        return t, prob(*t)

    def cpu(budget_s):
        from oracle import binding  # the checker: timed baseline and cross-check only
        port = binding.load_port()
        cores = _cpu_cores()
        a_h = a.view(torch.int16).cpu().numpy().view(np.uint16)
        b_h = b.cpu().numpy()
        s_h = s.view(torch.int16).cpu().numpy().view(np.uint16)

        def run_rows(r):
            return port.gemm_bf16_i8(np.ascontiguousarray(a_h[:r]), b_h, s_h, r, N, K, K, 1, K, N, 0, nthreads=cores)

        rows, dt, ref = _row_block(run_rows, M, budget_s)
        ref = (ref.reshape(rows, N).astype(np.uint32) << 16).view(np.float32)
        got = (c[:rows].view(torch.int16).cpu().numpy().view(np.uint16).astype(np.uint32) << 16).view(np.float32)
        err = np.abs(got - ref)
        # bf16 output: one truncation step (2^-7 relative) on top of the fp32 summation-order difference
        ok = bool((err <= 1e-2 * np.abs(ref) + 1e-2).all())
        return {"value": 2.0 * rows * N * K / dt / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
                "sample": f"rows 0..{rows - 1} of A x all N={N} x full K={K} ({rows}/{M} of the GEMM), {dt:.2f} s; "
                          f"full-size estimate {dt * M / rows:.0f} s",
                "gpu_rows_match_checker": ok, "max_abs_err": float(err.max())}

    return {"M": M, "N": N, "K": K, "problem": prob(a, bpre, c), "keep": (a, b, bpre, s, c), "clone": clone, "cpu": cpu,
            "dtype": "bf16", "peak": PEAK_FP16_MFMA_TFLOPS, "bytes": 2 * M * K + K * N + 2 * N + 2 * M * N,
            "workspace": None,
            "workload": "bf16A x int8B GEMM M=4096 N=11008 K=4096, per-channel bf16 scale (Multiply on B), bf16 C "
                        "(example/60_gemm_multi_ABD arithmetic, BASELINE config 3); int8 weights in the pre-shuffled order"}


def workload_c5(dev):
    """BASELINE config 5: fp8 A x packed-int4 B + MultiplyMultiply + FastGelu, 4096^3.  A holds E4M3 FNUZ bytes of
    moderate magnitude (no NaN / +-240 codes), the input contract of the fp8 matrix-core kernel."""
    M = N = K = 4096
    g = torch.Generator(device=dev)
    g.manual_seed(11939)
    a = (torch.randint(24, 88, (M, K), generator=g, device=dev, dtype=torch.int16) +
         128 * torch.randint(0, 2, (M, K), generator=g, device=dev, dtype=torch.int16)).to(torch.uint8)
    q = torch.randint(-8, 8, (N, K), generator=g, device=dev, dtype=torch.int8)
    bp = torch.empty((N, K // 2), dtype=torch.uint8, device=dev)
    capi.pack_int4(q.data_ptr(), K, K, N, bp.data_ptr(), K // 2, torch.cuda.current_stream().cuda_stream)
    bp = preshuffled(bp, K, N)
    d0 = torch.rand((N,), generator=g, device=dev) * 2.0 ** -4
    d1 = torch.rand((M,), generator=g, device=dev)
    c = torch.empty((M, N), dtype=torch.float16, device=dev)
    torch.cuda.synchronize()

    def prob(a_, b_, c_):
        return capi.fpAintB_problem(a_.data_ptr(), b_.data_ptr(), None, c_.data_ptr(), M, N, K, a_dtype=capi.F8_FNUZ,
                                    b_dtype=capi.I4_PRESHUFFLED, c_dtype=capi.F16, epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU,
                                    d0=d0.data_ptr(), d1=d1.data_ptr(), flags=capi.FLAG_F8_NO_SPECIALS)

    def clone():
        t = (a.clone(), bp.clone(), torch.empty_like(c))
        return t, prob(*t)

    def cpu(budget_s):
        from oracle import binding  # the checker: timed baseline and cross-check only
        port = binding.load_port()
        cores = _cpu_cores()
        a_h, q_h = a.cpu().numpy(), q.cpu().numpy()
        d0_h, d1_h = d0.cpu().numpy(), d1.cpu().numpy()

        def run_rows(r):
            return port.gemm_f8_i8_mm(np.ascontiguousarray(a_h[:r]), q_h, d0_h, np.ascontiguousarray(d1_h[:r]), r, N,
                                      K, K, K, N, True, nthreads=cores)

        rows, dt, ref = _row_block(run_rows, M, budget_s)
        ref = ref.reshape(rows, N).view(np.float16).astype(np.float32)
        got = c[:rows].cpu().numpy().view(np.float16).astype(np.float32)
        err = np.abs(got - ref)
        ok = bool((err <= 1e-3 + 1e-3 * np.abs(ref)).all())
        return {"value": 2.0 * rows * N * K / dt / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
                "sample": f"rows 0..{rows - 1} of A x all N={N} x full K={K} ({rows}/{M} of the GEMM), {dt:.2f} s; "
                          f"full-size estimate {dt * M / rows:.0f} s",
                "gpu_rows_match_checker": ok, "max_abs_err": float(err.max())}

    return {"M": M, "N": N, "K": K, "problem": prob(a, bp, c), "keep": (a, q, bp, d0, d1, c), "clone": clone,
            "cpu": cpu, "dtype": "f8", "peak": PEAK_8BIT_MFMA_TFLOPS,
            "bytes": M * K + K * N // 2 + 4 * (M + N) + 2 * M * N, "workspace": None,
            "workload": "fp8(E4M3 FNUZ)A x int4B GEMM M=N=K=4096 + MultiplyMultiply(d0[n], d1[m]) + FastGelu, fp16 C "
                        "(example/65_gemm_multiply_multiply arithmetic, BASELINE config 5); int4 weights in the pre-shuffled order"}


WORKLOADS = {"C1": workload_c1, "C2": workload_c2, "C3": workload_c3, "C5": workload_c5}
HEADLINE_METRIC = "TFLOPs + achieved HBM GB/s, fp16A x int4B GEMM at Llama-7B shapes, 1 GPU"
METRICS = {"C2": HEADLINE_METRIC,
           "C1": "TOPs + achieved HBM GB/s, int8 x int8 requantised GEMM 256^3 (BASELINE config 1), 1 GPU",
           "C3": "TFLOPs + achieved HBM GB/s, bf16A x int8B GEMM at the Llama-7B FFN shape (BASELINE config 3), 1 GPU",
           "C5": "TFLOPs + achieved HBM GB/s, fp8A x int4B GEMM + MultiplyMultiply + FastGelu (BASELINE config 5), 1 GPU"}


def run_single(args):
    dev = torch.device("cuda", 0)
    w = WORKLOADS[args.config](dev)
    M, N, K, p = w["M"], w["N"], w["K"], w["problem"]
    assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    # untimed warm-up: the W requested steps plus enough launches (~0.2 s) for the clocks to settle
    ramp = max(0, args.ramp_launches)
    for _ in range(args.warmup + ramp):
        capi.launch(p, None, sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        capi.launch(p, None, sp)
    e1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / args.steps
    flops = 2.0 * M * N * K
    tflops = flops * args.steps / wall / 1e12
    kern_tflops = flops / (kern_ms * 1e-3) / 1e12
    nbytes = w["bytes"]
    traffic, traffic_source = pmc_traffic(capi.kernel_name(p), args.config)
    out = {
        "metric": METRICS[args.config],
        "value": tflops,
        "unit": "TFLOP/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "warmup_extra_untimed_launches": ramp,
        "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": w["dtype"],
        "data": "synthetic",
        "config": {"workload": w["workload"], "baseline_config": args.config,
                   "M": M, "N": N, "K": K, "kernel": capi.kernel_name(p), "instance": capi.type_string(p),
                   # the order the weight bytes are stored in: a one-off offline re-layout of the packed / int8 matrix
                   # (same bytes, same output bits); the plain order is timed beside it for C2 (packed_weights_tflops)
                   "weights_layout": {capi.I4_PRESHUFFLED: "I4_PRESHUFFLED", capi.I8_PRESHUFFLED: "I8_PRESHUFFLED",
                                      capi.I4_PACKED: "I4_PACKED", capi.I8: "I8"}.get(p.b_dtype, str(p.b_dtype))},
        "hbm_gbps_algorithmic": nbytes / (kern_ms * 1e-3) / 1e9,
        "hbm_frac_of_peak": nbytes / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
        "roofline": {"bound": "mfma", "achieved": kern_tflops, "peak": w["peak"],
                     "unit": "TFLOP/s", "frac": kern_tflops / w["peak"],
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_ms": kern_ms,
                     "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes},
    }
    # Outside the timed region: per-launch times (median next to the mean above) and a rotating-buffer pass in
    # the manner of the reference's flush mode (include/ck/host_utility/flush_cache.hpp:107-235): 4 copies of every
    # large operand (more than the 256 MiB last-level cache at the 4096-class sizes) used round-robin, so no launch
    # finds its inputs cached by the previous one.
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(51)]
    evs[0].record(stream)
    for i in range(50):
        capi.launch(p, None, sp)
        evs[i + 1].record(stream)
    torch.cuda.synchronize()
    per_launch = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(50))
    out["roofline"]["kernel_ms_median"] = per_launch[25]
    copies = [w["clone"]() for _ in range(4)]
    probs = [c[1] for c in copies]
    for i in range(40):
        capi.launch(probs[i % 4], None, sp)
    torch.cuda.synchronize()
    e0.record(stream)
    for i in range(100):
        capi.launch(probs[i % 4], None, sp)
    e1.record(stream)
    torch.cuda.synchronize()
    rot_ms = e0.elapsed_time(e1) / 100
    out["rotating_buffers"] = {"copies": 4, "kernel_ms": rot_ms, "tflops": flops / (rot_ms * 1e-3) / 1e12}
    del copies, probs
    if args.config == "C2":
        # Secondary figure, never `value`: the same GEMM with the per-channel scale applied once in the epilogue
        # (scale = NULL, WQ_EPI_MULTIPLY with d0 = scale) instead of per weight in the loop.  It rounds differently from
        # the reference (which rounds half(q) * scale per weight: reference_fpAintB_gemm.hpp:110) -- more accurately,
        # within check_err's tolerance -- so it is the caller's choice, not the measured configuration.
        a_, q_, s_, bp_, c_, _bpre = w["keep"]
        pf = capi.fpAintB_problem(a_.data_ptr(), bp_.data_ptr(), None, c_.data_ptr(), M, N, K, epilogue=capi.EPI_MULTIPLY,
                                  d0=s_.data_ptr())
        for _ in range(50):
            capi.launch(pf, None, sp)
        e0.record(stream)
        for _ in range(100):
            capi.launch(pf, None, sp)
        e1.record(stream)
        torch.cuda.synchronize()
        out["fast_mode_tflops"] = flops / (e0.elapsed_time(e1) / 100 * 1e-3) / 1e12
        # Secondary figure, never `value`: the measured configuration with the same weights in the plain packed order
        # (WQ_I4_PACKED, column after column): same kernel and arithmetic, same output bits (compared below); the
        # difference is how well a wave-wide weight fetch coalesces.
        c2_ = torch.empty_like(c_)
        pk = capi.fpAintB_problem(a_.data_ptr(), bp_.data_ptr(), s_.data_ptr(), c2_.data_ptr(), M, N, K)
        for _ in range(50):
            capi.launch(pk, None, sp)
        e0.record(stream)
        for _ in range(100):
            capi.launch(pk, None, sp)
        e1.record(stream)
        torch.cuda.synchronize()
        out["packed_weights_tflops"] = flops / (e0.elapsed_time(e1) / 100 * 1e-3) / 1e12
        capi.launch(p, None, sp)  # leave C as the measured configuration computes it (the CPU leg compares it)
        torch.cuda.synchronize()
        out["packed_weights_same_bits"] = bool(torch.equal(c_, c2_))
    if not args.no_cpu:
        out["cpu_baseline"] = w["cpu"](args.cpu_budget)
    print(json.dumps(out))


def run_splitk(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    have = torch.cuda.device_count()
    if local >= have:
        raise SystemExit(f"bench.py --gpus {world}: rank {rank} needs GPU {local}, this node shows {have}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    M = N = 4096
    K = args.splitk_k
    if M % world != 0:
        raise SystemExit(f"bench.py --gpus {world}: M = {M} must be divisible by the number of GPUs")
    sl = splitk.partition(K, world)[rank]
    klen = sl.length
    # each rank owns its K slice of A and of the packed weights (synthetic, per-rank seed)
    a, q, s, bp = synth_inputs(M, N, klen, 11939 + rank, dev)
    # the rank's slice of the weights in the pre-shuffled order (as a deployment would store it; same output bits)
    b_dtype = capi.I4_PACKED
    if klen > 0 and klen % 64 == 0:
        bp, b_dtype = preshuffled(bp, klen, N), capi.I4_PRESHUFFLED
    s = s.clone()
    dist.broadcast(s, src=0)  # the per-channel scales are replicated
    partial = torch.empty((M, N), dtype=torch.float32, device=dev)
    rows = M // world
    c_full = torch.empty((M, N), dtype=torch.float16, device=dev)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream

    # ---- the step, form 1 (the product path): wqgemm_splitk_multi_step of the C ABI -- slice GEMM, ncclReduceScatter,
    # finalize of the rank's rows, ncclAllGather, all issued by libwqgemm.so on this stream through an ncclComm_t made
    # with RCCL's own C API (unique id from rank 0, handed round over the process group).
    exchange, comm, comm_error = "c_abi_rccl", None, None
    if args.exchange == "torch":
        exchange = "torch_distributed"
    else:
        # Every rank takes part in the broadcast whatever happens before it (a rank that raised ahead of a collective
        # would leave the others waiting in it): rank 0 sends zeros when it could not make an id, and nobody then
        # calls ncclCommInitRank.
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        rccl = None
        try:
            from composable_kernel_amd import rccl
            assert rccl.UNIQUE_ID_BYTES == uid.numel()
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(rccl.get_unique_id()), dtype=torch.uint8))
        except Exception as exc:  # noqa: BLE001  (reported in the JSON line; the torch.distributed form then runs)
            exchange, comm_error, rccl = "torch_distributed", f"{type(exc).__name__}: {exc}", None
            uid.zero_()
        dist.broadcast(uid, src=0)
        have_id = bool(uid.any().item())
        can = torch.tensor([1 if (rccl is not None and have_id) else 0], device=dev)
        dist.all_reduce(can, op=dist.ReduceOp.MIN)  # (ncclCommInitRank is itself collective: all ranks or none)
        if int(can.item()) == 1:
            try:
                comm = rccl.comm_init_rank(world, bytes(uid.cpu().numpy().tobytes()), rank)
            except Exception as exc:  # noqa: BLE001
                exchange, comm_error = "torch_distributed", f"{type(exc).__name__}: {exc}"
        elif exchange == "c_abi_rccl":
            exchange, comm_error = "torch_distributed", comm_error or "another rank could not load RCCL's C API"
    ok_all = torch.tensor([1 if comm is not None else 0], device=dev)
    dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
    if int(ok_all.item()) == 0 and exchange == "c_abi_rccl":
        exchange = "torch_distributed"  # (some rank could not make its communicator)

    def multi_problem(a_t, s_t):
        return capi.fpAintB_problem(a_t.data_ptr(), bp.data_ptr(), s_t.data_ptr(), c_full.data_ptr(), M, N, klen,
                                    lda=max(klen, 1), ldb=max(klen, 64), b_dtype=b_dtype)

    p_multi = multi_problem(a, s)
    ws_multi = torch.empty(max(capi.splitk_multi_workspace_bytes(p_multi, world), 1), dtype=torch.uint8, device=dev)

    # ---- the step, form 2: the same flow with torch.distributed's collectives (splitk.splitk_gemm_step, the function
    # the gloo tests run on CPU); timed as well, and what runs if the communicator above could not be made.
    def slice_problem(a_t, s_t):
        pg = capi.fpAintB_problem(a_t.data_ptr(), bp.data_ptr(), s_t.data_ptr(), partial.data_ptr(), M, N, max(klen, 64),
                                  k_begin=0, k_end=max(klen, 64), c_dtype=capi.F32, b_dtype=b_dtype)
        assert klen == 0 or capi.is_supported(pg), capi.lib.wqgemm_last_error().decode()
        return pg

    p_gemm = slice_problem(a, s)
    cur = {"gemm": p_gemm, "scale": s, "multi": p_multi}

    def compute_partial(sl, partial_):
        if sl.length == 0:
            partial_.zero_()  # a rank without a slice contributes zeros
        else:
            capi.launch(cur["gemm"], None, sp)

    def finalize(summed, out_rows, row0):
        # PassThrough epilogue: no D operand, so the row offset only selects where the rows land (out_rows)
        pf = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), cur["scale"].data_ptr(), out_rows.data_ptr(),
                                  summed.shape[0], N, max(klen, 64), b_dtype=b_dtype)
        capi.reduce_finalize(pf, summed.data_ptr(), 1, sp)

    def step_torch():
        splitk.splitk_gemm_step(compute_partial, finalize, partial, c_full, K)

    def step_cabi():
        capi.splitk_multi_step(cur["multi"], comm, rank, world, ws_multi.data_ptr(), sp)

    step = step_cabi if exchange == "c_abi_rccl" else step_torch

    # ---- untimed check of the whole flow on exact (integer-valued) inputs: rank 0 recomputes a block of C from the
    # operands of every rank in float64 and compares bit for bit (all partial sums are exact in fp32).
    check = None
    if klen > 0 and all(sl_.length == klen for sl_ in splitk.partition(K, world)):
        g = torch.Generator(device=dev)
        g.manual_seed(777 + rank)
        a_int = torch.randint(-2, 3, (M, klen), generator=g, device=dev, dtype=torch.int8).to(torch.float16)
        s_int = torch.randint(1, 3, (N,), generator=torch.Generator(device=dev).manual_seed(5), device=dev,
                              dtype=torch.int8).to(torch.float16)
        cur["gemm"], cur["scale"], cur["multi"] = slice_problem(a_int, s_int), s_int, multi_problem(a_int, s_int)
        R, C = 8, 64
        a_blk = torch.empty((world, R, klen), dtype=torch.float16, device=dev)
        q_blk = torch.empty((world, C, klen), dtype=torch.int8, device=dev)
        if world > 1:
            dist.all_gather_into_tensor(a_blk, a_int[:R].contiguous())
            dist.all_gather_into_tensor(q_blk, q[:C].contiguous())
        else:
            a_blk[0], q_blk[0] = a_int[:R], q[:C]
        exp = None
        if rank == 0:
            exp = torch.zeros((R, C), dtype=torch.float64)
            for r in range(world):
                exp += a_blk[r].cpu().double() @ (q_blk[r].cpu().double() * s_int[:C].cpu().double()[:, None]).T
            check = {"rows": R, "cols": C, "max_abs": float(exp.abs().max())}
        forms = [("c_abi_rccl", step_cabi)] if comm is not None and exchange == "c_abi_rccl" else []
        forms.append(("torch_distributed", step_torch))
        for name, fn in forms:
            c_full.zero_()
            fn()
            torch.cuda.synchronize()
            # (every rank holds the same C after the all-gather: compare rank 0's copy with the last rank's)
            probe_c = c_full[M - 1, :C].clone()
            if world > 1:
                dist.broadcast(probe_c, src=world - 1)
            if rank == 0:
                check[name] = {"exact_match": bool(torch.equal(c_full[:R, :C].cpu(), exp.to(torch.float16))),
                               "ranks_agree": bool(torch.equal(probe_c, c_full[M - 1, :C]))}
        if rank == 0:
            timed = check[exchange]
            check["exact_match"], check["ranks_agree"] = timed["exact_match"], timed["ranks_agree"]
        cur["gemm"], cur["scale"], cur["multi"] = p_gemm, s, p_multi
        del a_int, a_blk, q_blk

    for _ in range(args.warmup):
        step()
    # further untimed steps (~0.2 s, the same count on every rank) so that the GPU clocks have settled, as in the
    # single-GPU run
    torch.cuda.synchronize()
    t_probe = time.perf_counter()
    step()
    torch.cuda.synchronize()
    probe = torch.tensor([time.perf_counter() - t_probe], dtype=torch.float64, device=dev)
    dist.all_reduce(probe, op=dist.ReduceOp.MAX)
    extra = int(max(0, min(200, 0.2 / max(float(probe.item()), 1e-5)))) if args.ramp_launches > 0 else 0
    for _ in range(extra):
        step()
    torch.cuda.synchronize()
    # the timed region: barrier + synchronize on both sides of exactly `steps` steps.  The closing barrier (a small
    # collective plus a host round trip) is inside the region and is itemised below as closing_barrier_total_ms.
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dist.barrier()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    times = torch.tensor([t2 - t0, t1 - t0, e0.elapsed_time(e1) * 1e-3], dtype=torch.float64, device=dev)
    dist.all_reduce(times, op=dist.ReduceOp.MAX)
    wall, wall_steps, gpu_steps = (float(v) for v in times.tolist())

    # the other form of the step, timed the same way outside the headline region (for the record)
    other_ms = None
    if exchange == "c_abi_rccl":
        n_other = max(1, min(args.steps, 20))
        for _ in range(2):
            step_torch()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(n_other):
            step_torch()
        e1.record(stream)
        torch.cuda.synchronize()
        t_o = torch.tensor([e0.elapsed_time(e1) / n_other], dtype=torch.float64, device=dev)
        dist.all_reduce(t_o, op=dist.ReduceOp.MAX)
        other_ms = float(t_o.item())

    # itemise one step with events (outside the timed region): the four parts, issued separately
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    n_probe = 5
    acc = np.zeros(4)
    for _ in range(n_probe):
        ev[0].record(stream)
        compute_partial(sl, partial)
        ev[1].record(stream)
        summed = splitk.reduce_partials(partial, "reduce_scatter") if world > 1 else partial[rank * rows:(rank + 1) * rows]
        ev[2].record(stream)
        finalize(summed, c_full[rank * rows:(rank + 1) * rows], rank * rows)
        ev[3].record(stream)
        splitk.gather_rows(c_full, rows)
        ev[4].record(stream)
        torch.cuda.synchronize()
        acc += np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
    parts = torch.tensor(acc / n_probe, dtype=torch.float64, device=dev)
    dist.all_reduce(parts, op=dist.ReduceOp.MAX)
    gemm_ms, reduce_ms, fin_ms, gather_ms = (float(v) for v in parts.tolist())

    if rank == 0:
        flops_total = 2.0 * M * N * K
        flops_rank = 2.0 * M * N * klen
        kern_tflops = flops_rank / (gemm_ms * 1e-3) / 1e12
        out = {
            "metric": HEADLINE_METRIC,
            "value": flops_total * args.steps / wall / 1e12,
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_extra_untimed_steps": extra + 1,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"fp16A x int4B split-K GEMM M=N=4096 K={K} across {world} GPUs, "
                                   "fp32 partials, RCCL reduce-scatter + fp16 all-gather (BASELINE config 4)",
                       "M": M, "N": N, "K": K, "k_per_gpu": klen, "parallelism": f"splitk{world}",
                       "kernel": capi.kernel_name(p_gemm), "weights_layout": "I4_PRESHUFFLED" if b_dtype == capi.I4_PRESHUFFLED else "I4_PACKED"},
            "exchange": {"timed": exchange,
                         "c_abi_rccl": "wqgemm_splitk_multi_step (libwqgemm.so issues ncclReduceScatter / ncclAllGather on an ncclComm_t)",
                         "torch_distributed": "splitk.splitk_gemm_step (reduce_scatter_tensor / all_gather_into_tensor)",
                         "communicator_error": comm_error,
                         "other_form_gpu_ms_per_step": other_ms},
            "splitk_check": check,
            # wall = steps (host-timed to the end of the last step's GPU work) + the closing barrier; the GPU-side
            # figure is the same steps between two events on the launch stream
            "timing": {"steps_only_ms_per_step": wall_steps * 1e3 / args.steps,
                       "gpu_events_ms_per_step": gpu_steps * 1e3 / args.steps,
                       "closing_barrier_total_ms": (wall - wall_steps) * 1e3},
            "breakdown_ms": {"gemm": gemm_ms, "reduce_scatter_fp32": reduce_ms, "finalize": fin_ms,
                             "all_gather_fp16": gather_ms, "sum_of_parts": gemm_ms + reduce_ms + fin_ms + gather_ms,
                             "note": "the four parts issued one by one through torch.distributed, mean of 5"},
            "roofline": {"bound": "mfma", "achieved": kern_tflops, "peak": PEAK_FP16_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": kern_tflops / PEAK_FP16_MFMA_TFLOPS, "traffic": None,
                         "traffic_source": None, "kernel_ms": gemm_ms, "algorithmic_flops_per_launch": flops_rank,
                         "algorithmic_bytes_per_launch": algorithmic_bytes(M, N, klen, c_bytes=4)},
        }
        print(json.dumps(out), flush=True)
    if comm is not None:
        from composable_kernel_amd import rccl
        torch.cuda.synchronize()
        rccl.comm_destroy(comm)
    dist.destroy_process_group()


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n_gpus, argv, device_count=None, run=None, out=sys.stdout, err=sys.stderr) -> int:
    """`bench.py --gpus N` started by hand (no RANK / WORLD_SIZE in the environment): start the N ranks as fresh child
    processes -- one per GPU, through torch.distributed.run on 127.0.0.1 -- BEFORE this process makes any GPU call, pass
    rank 0's JSON line through to stdout and everything else to stderr, and return the children's exit status.
    device_count / run are injection points for the CPU test."""
    import subprocess
    have = (device_count or torch.cuda.device_count)()  # counts devices without initialising the GPU runtime
    if have < n_gpus:
        print(f"bench.py --gpus {n_gpus}: needs {n_gpus} GPUs, found {have}", file=err)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver shares device memory through dmabuf only)
    r = (run or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = 0
    for line in (r.stdout or "").splitlines():
        is_result = False
        if line.startswith("{"):
            try:
                is_result = "metric" in json.loads(line)
            except ValueError:
                pass
        if is_result and lines == 0:
            print(line, file=out, flush=True)
            lines += 1
        else:
            print(line, file=err)
    if r.stderr:
        print(r.stderr, file=err, end="")
    if r.returncode == 0 and lines != 1:
        print(f"bench.py --gpus {n_gpus}: the ranks exited cleanly without a result line", file=err)
        return 1
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for the baseline")
    ap.add_argument("--splitk-k", type=int, default=65536)
    ap.add_argument("--config", choices=("C1", "C2", "C3", "C4", "C5"), default="C2",
                    help="1-GPU workload: C2 (default) is the configuration the metric is quoted on; C1, C3, C5 are the "
                         "remaining single-GPU BASELINE configurations, each with its own CPU baseline; C4 runs the "
                         "multi-GPU split-K flow (what --gpus N > 1 runs) on --gpus ranks, also on one")
    ap.add_argument("--ramp-launches", type=int, default=1500,
                    help="extra untimed launches before the timed region so the GPU clocks have settled")
    ap.add_argument("--exchange", choices=("c_abi", "torch"), default="c_abi",
                    help="N > 1: which form of the split-K step is timed -- the C ABI's wqgemm_splitk_multi_step over an "
                         "ncclComm_t (default) or the same flow through torch.distributed's collectives")
    args = ap.parse_args()
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if (args.gpus > 1 or args.config == "C4") and not launched:
        # started by hand: this process only starts the ranks (no GPU call is made here, before or after)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (run it through gpurun)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or args.config == "C4":
        run_splitk(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
