#!/usr/bin/env python3
"""A few mid-sized-M problems launched 200 times each with the planner's optional workspace: run under
`rocprofv3 --kernel-trace --stats` to see which kernels the planner's split-K routes consist of and what each costs.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mid_m -- python3 tools/mid_m_case.py
"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from composable_kernel_amd import capi

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
sp = torch.cuda.current_stream().cuda_stream
for kind, M, N, K in (("f16", 768, 4096, 11008), ("f8", 96, 4096, 11008), ("i8", 96, 4096, 11008)):
    if kind == "i8":
        a = torch.randint(-128, 128, (M, K), generator=g, device=dev, dtype=torch.int8)
        w = torch.randint(-128, 128, (N, K), generator=g, device=dev, dtype=torch.int8)
        c = torch.empty((M, N), dtype=torch.int8, device=dev)
        p = capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.I8, b_dtype=capi.I8,
                                 c_dtype=capi.I8, epilogue=capi.EPI_ACT_MUL_CLAMP, requant=0.01)
    else:
        w = torch.randint(0, 256, (N, K // 2), generator=g, device=dev, dtype=torch.uint8)
        c = torch.empty((M, N), dtype=torch.float16, device=dev)
        if kind == "f8":
            a = torch.randint(0, 120, (M, K), generator=g, device=dev, dtype=torch.uint8)
            d0 = torch.rand((N,), generator=g, device=dev); d1 = torch.rand((M,), generator=g, device=dev)
            p = capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16,
                                     b_dtype=capi.I4_PRESHUFFLED, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0.data_ptr(), d1=d1.data_ptr())
        else:
            a = (torch.rand((M, K), generator=g, device=dev) * 2 - 1).half()
            s = (torch.rand((N,), generator=g, device=dev) * 0.01 + 0.001).half()
            p = capi.fpAintB_problem(a.data_ptr(), w.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)
    ws = torch.empty(max(capi.workspace_bytes(p), 16), dtype=torch.uint8, device=dev)
    print(kind, M, N, K, capi.kernel_name(p), capi.type_string(p), flush=True)
    for _ in range(200):
        capi.launch(p, ws.data_ptr(), sp)
    torch.cuda.synchronize()
