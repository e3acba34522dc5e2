#!/usr/bin/env python3
"""Mid-sized M (prefill chunks): the planner's own choice against every forced tile size and k_batch, per shape.

    python tools/mid_m_sweep.py [--pre] > gpurun_out/mid_m.jsonl

One JSON line per (N, K, M): microseconds of the planner's choice (with a workspace), of each candidate
(tile, k_batch), and the ratio planner / best.
"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.getcwd())
from composable_kernel_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--pre", action="store_true", help="pre-shuffled weights")
ap.add_argument("--b", choices=("i4", "i8"), default="i4")
ap.add_argument("--i8a", action="store_true", help="int8 activations x int8 weights, requantised int8 output (no split-K form)")
ap.add_argument("--f8", action="store_true", help="fp8 (E4M3FN) activations, MultiplyMultiply epilogue: the fp8 matrix-core kernel")
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
stream = torch.cuda.current_stream(); sp = stream.cuda_stream
bits4 = args.b == "i4" and not args.i8a
b_dtype = (capi.I4_PRESHUFFLED if args.pre else capi.I4_PACKED) if bits4 else (capi.I8_PRESHUFFLED if args.pre else capi.I8)
for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (8192, 8192), (28672, 8192)):
    w = torch.randint(0, 256, (N, K // 2 if bits4 else K), generator=g, device=dev, dtype=torch.uint8)   # timing only: any bytes
    s = (torch.rand((N,), generator=g, device=dev) * 0.01 + 0.001).half()
    for M in ((1, 16, 64) if args.f8 else ()) + (96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 3072):
        a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).half()
        if args.f8:
            a = torch.randint(0, 120, (M, K), generator=g, device=dev, dtype=torch.uint8)   # E4M3FN bytes below NaN
            d0 = torch.rand((N,), generator=g, device=dev)
            d1 = torch.rand((M,), generator=g, device=dev)
        c = torch.empty((M, N), dtype=torch.float16, device=dev)
        if args.i8a:
            a = torch.randint(-128, 128, (M, K), generator=g, device=dev, dtype=torch.int8)

        def timed(flags, kb):
            if args.i8a:
                if kb != 1:
                    return None, None
                p = capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.I8, b_dtype=capi.I8,
                                         c_dtype=capi.I8, epilogue=capi.EPI_ACT_MUL_CLAMP, requant=0.01, flags=flags)
            elif args.f8:
                p = capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_E4M3FN,
                                         c_dtype=capi.F16, b_dtype=b_dtype, k_batch=kb, flags=flags,
                                         epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0.data_ptr(), d1=d1.data_ptr())
            else:
                p = capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, b_dtype=b_dtype,
                                         k_batch=kb, flags=flags)
            if not capi.is_supported(p):
                return None, None
            nb = capi.workspace_bytes(p)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            wsp = ws.data_ptr() if nb else None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def run(n):
                e0.record(stream)
                for _ in range(n):
                    capi.launch(p, wsp, sp)
                e1.record(stream); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n * 1e3
            run(30)
            return round(statistics.median(run(30) for _ in range(5)), 1), capi.type_string(p)

        planner_us, planner_ts = timed(0, 1)
        cands = {}
        for tname, flag in ((("f8", 0),) if args.f8 else (("t128", capi.FLAG_TILE_128), ("t256", capi.FLAG_TILE_256))):
            for kb in (1, 2, 3, 4, 6, 8):
                if K // kb < 512:
                    continue
                us, _ = timed(flag, kb)
                if us is not None:
                    cands[f"{tname}_kb{kb}"] = us
        best = min(cands, key=cands.get)
        print(json.dumps(dict(N=N, K=K, M=M, planner_us=planner_us, planner=planner_ts, best=best, best_us=cands[best],
                              planner_over_best=round(planner_us / cands[best], 3), candidates=cands)), flush=True)
