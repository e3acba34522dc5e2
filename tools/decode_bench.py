#!/usr/bin/env python3
"""Small-M (decode) shapes: time, algorithmic HBM GB/s and fraction of the 8 TB/s HBM roofline.

    python tools/decode_bench.py [--out profiles/r1_decode.jsonl] [--b i4|i8]

Weights are rotated over enough copies to exceed the 256 MiB last-level cache, so every launch streams its weights from
HBM.  Timing: median of 5 x 40 launches between events after a short clock-settling run.
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from composable_kernel_amd import capi  # noqa: E402

PEAK_HBM = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--b", default="i4", choices=("i4", "i8"))
    ap.add_argument("--a", default="f16", choices=("f16", "i8"), help="i8: the int8 x int8 requantised GEMM (int8 weights and output)")
    ap.add_argument("--flags", type=int, default=0, help="planner flags (e.g. 1 = the 128-tile kernel) for comparisons")
    ap.add_argument("--group-k", type=int, default=0, help="group-wise weight scales: k per group (fp16 activations)")
    ap.add_argument("--pre", action="store_true", help="weights in the pre-shuffled order (random bytes either way)")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008,8192x8192,28672x8192,8192x28672")
    ap.add_argument("--ms", default="1,4,16,32,64")
    ap.add_argument("--graph", action="store_true",
                    help="time replays of a HIP graph holding 40 launches (no per-launch host work) instead of plain launches")
    args = ap.parse_args()
    if args.a == "i8":
        args.b = "i8"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    lines = []
    # what a plain streaming pass reaches on this box: device-to-device copy (reads + writes) and a sum (reads only) over 1 GiB
    x = torch.empty(1 << 28, dtype=torch.int32, device=dev).random_(generator=g)
    y = torch.empty_like(x)
    for name, fn, nbytes in (("copy 1 GiB (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4),
                             ("sum 1 GiB (read)", lambda: x.sum(), x.numel() * 4)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            fn()
        e0.record(stream)
        for _ in range(10):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line = {"probe": name, "ms": ms, "hbm_gbps": nbytes / (ms * 1e-3) / 1e9}
        print(json.dumps(line), flush=True)
        lines.append(line)
    del x, y
    for shape in args.shapes.split(","):
        N, K = (int(v) for v in shape.split("x"))
        wbytes = N * K // 2 if args.b == "i4" else N * K
        copies = max(2, min(16, (320 << 20) // wbytes + 1))
        if args.b == "i4":
            ws = [torch.randint(0, 256, (N, K // 2), generator=g, device=dev, dtype=torch.uint8) for _ in range(copies)]
            bdt = capi.I4_PRESHUFFLED if args.pre else capi.I4_PACKED
        else:
            ws = [torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8) for _ in range(copies)]
            bdt = capi.I8_PRESHUFFLED if args.pre else capi.I8
        s = (torch.rand((K // args.group_k, N) if args.group_k else (N,), generator=g, device=dev) * 0.01 + 0.001).half()
        for M in (int(v) for v in args.ms.split(",")):
            if args.a == "i8":
                a = torch.randint(-128, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
                c = torch.empty((M, N), dtype=torch.int8, device=dev)
                probs = [capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.I8, b_dtype=capi.I8,
                                              c_dtype=capi.I8, epilogue=capi.EPI_ACT_MUL_CLAMP, requant=1e-3, flags=args.flags)
                         for w in ws]
            else:
                a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).half()
                c = torch.empty((M, N), dtype=torch.float16, device=dev)
                probs = [capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, b_dtype=bdt,
                                              scale_group_k=args.group_k, flags=args.flags)
                         for w in ws]
            nb = max(capi.workspace_bytes(p) for p in probs)
            wsp = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            graph = None
            if args.graph:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    cs = torch.cuda.current_stream().cuda_stream
                    for i in range(40):
                        capi.launch(probs[i % copies], wsp.data_ptr() if nb else None, cs)

            def run(n):
                e0.record(stream)
                if graph is not None:
                    for _ in range(max(1, n // 40)):
                        graph.replay()
                    n = max(1, n // 40) * 40
                else:
                    for i in range(n):
                        capi.launch(probs[i % copies], wsp.data_ptr() if nb else None, sp)
                e1.record(stream)
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n

            run(200)
            ms = statistics.median(run(40) for _ in range(5))
            nbytes = wbytes + M * K + M * N if args.a == "i8" else wbytes + 2 * M * K + 2 * M * N + 2 * N
            line = {"N": N, "K": K, "M": M, "a": args.a, "workspace_bytes": nb, "b": args.b + ("_preshuffled" if args.pre else ""), "kernel": capi.kernel_name(probs[0]), "us": ms * 1e3,
                    "algorithmic_bytes": nbytes, "hbm_gbps": nbytes / (ms * 1e-3) / 1e9,
                    "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM, "weight_copies_rotated": copies,
                    "hip_graph": bool(args.graph), "scale_group_k": args.group_k}
            print(json.dumps(line), flush=True)
            lines.append(line)
        del ws
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            for line in lines:
                fh.write(json.dumps(line) + "\n")


if __name__ == "__main__":
    main()
