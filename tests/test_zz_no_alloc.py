"""The allocation-free entry point wqgemm_launch_noalloc against wqgemm_launch with a caller workspace: same bits,
on two calls in a row (the second smaller, so it sees the first call's leftovers in the shared scratch), plus the
size limit.  Everything goes through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

from composable_kernel_amd import capi
from gpu_util import dev, f16_bits, host_u16, pack_on_device

pytestmark = pytest.mark.gpu


def _case(rng, M, N, K, kb):
    a_t = dev(f16_bits(rng.uniform(-1, 1, (M, K))))
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    b_t = pack_on_device(q, K, N)
    s_t = dev(f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N)))
    outs = []
    st = torch.cuda.current_stream().cuda_stream
    for which in ("workspace", "noalloc"):
        c_t = torch.full((M, N), 0x7BFF, dtype=torch.int16, device="cuda")
        p = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K, k_batch=kb)
        assert capi.is_supported(p)
        need = capi.workspace_bytes(p)
        assert 0 < need <= capi.NOALLOC_SCRATCH_BYTES
        if which == "workspace":
            ws = torch.empty(need, dtype=torch.uint8, device="cuda")
            capi.launch(p, ws.data_ptr(), st)
        else:
            capi.launch_noalloc(p, st)
        torch.cuda.synchronize()
        outs.append(host_u16(c_t).copy())
    np.testing.assert_array_equal(outs[1], outs[0])
    assert (outs[0] != 0x7BFF).any()


def test_noalloc_matches_workspace_launch_twice_then_limit():
    rng = np.random.default_rng(2024)
    _case(rng, 512, 512, 2048, 4)      # 4 MiB of slabs
    _case(rng, 128, 256, 1024, 3)      # smaller, right after: the scratch still holds the first call's slabs
    # row-major 8-bit weights + split-K: slabs and the K-contiguous copy both live in the scratch
    M, N, K, kb = 256, 256, 512, 2
    a_t = dev(f16_bits(rng.uniform(-1, 1, (M, K))))
    brow = dev(rng.integers(-128, 128, (K, N)).astype(np.int8))
    s_t = dev(f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N)))
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for which in range(2):
        c_t = torch.zeros((M, N), dtype=torch.int16, device="cuda")
        p = capi.fpAintB_problem(a_t.data_ptr(), brow.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K, b_dtype=capi.I8,
                                 b_layout=capi.ROW_MAJOR, ldb=N, k_batch=kb)
        if which == 0:
            ws = torch.empty(capi.workspace_bytes(p), dtype=torch.uint8, device="cuda")
            capi.launch(p, ws.data_ptr(), st)
        else:
            capi.launch_noalloc(p, st)
        torch.cuda.synchronize()
        res.append(host_u16(c_t).copy())
    np.testing.assert_array_equal(res[1], res[0])
    # beyond the limit: refused with a message, nothing launched
    big = capi.fpAintB_problem(a_t.data_ptr(), brow.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), 4096, 4096, 4096,
                               k_batch=2)
    assert capi.workspace_bytes(big) > capi.NOALLOC_SCRATCH_BYTES
    assert capi.lib.wqgemm_launch_noalloc(C.byref(big), None) == capi.ERR_WORKSPACE
    assert b"built-in scratch" in capi.lib.wqgemm_last_error()
    torch.cuda.synchronize()
