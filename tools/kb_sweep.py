#!/usr/bin/env python3
"""Split-K sweep for mid-sized M on the 128-tile kernel: time per k_batch (0 = the planner's own choice with a
workspace, 1 = no workspace / one slice) -- the data behind the planner's rule for short grids.

    python tools/kb_sweep.py
"""
import sys, os, statistics, torch
sys.path.insert(0, os.getcwd())
from composable_kernel_amd import capi
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
stream = torch.cuda.current_stream(); sp = stream.cuda_stream
for (N, K) in ((4096, 4096), (11008, 4096), (8192, 8192)):
    w = torch.randint(0, 256, (N, K // 2), generator=g, device=dev, dtype=torch.uint8)
    s = (torch.rand((N,), generator=g, device=dev) * 0.01 + 0.001).half()
    for M in (192, 256, 384, 512, 768, 1024, 1536):
        a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).half()
        c = torch.empty((M, N), dtype=torch.float16, device=dev)
        res = []
        for kb in (0, 1, 2, 4):
            p = capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, k_batch=max(kb, 1))
            nb = capi.workspace_bytes(p)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            wsp = ws.data_ptr() if (nb and kb != 1) else None
            if kb == 1 and nb:   # planner's choice suppressed: no workspace -> one slice
                wsp = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            def run(n):
                e0.record(stream)
                for _ in range(n): capi.launch(p, wsp, sp)
                e1.record(stream); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n * 1e3
            run(100)
            res.append((kb, round(statistics.median(run(50) for _ in range(5)), 1), capi.type_string(p)[-9:-1] if kb == 0 else ""))
        print(N, K, M, res, flush=True)
