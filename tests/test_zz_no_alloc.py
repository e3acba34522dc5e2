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


def test_noalloc_optional_scratch_dropped_and_no_scratch_needed():
    """The two branches that use no scratch: an OPTIONAL workspace use larger than the built-in scratch is dropped exactly
    as wqgemm_launch drops it for a NULL workspace (here: the K-contiguous copy of row-major weights), and a problem that needs no workspace at all goes straight through.  Same bits as the plain launch."""
    rng = np.random.default_rng(77)
    st = torch.cuda.current_stream().cuda_stream
    # row-major 8-bit weights of 8192 x 8448 bytes: their K-contiguous copy (66 MiB) is an optional workspace use that does
    # not fit the 64 MiB scratch, so both entry points run the general kernel on the weights as they are
    M, N, K = 32, 8448, 8192
    a_t = dev(f16_bits(rng.integers(-2, 3, (M, K)).astype(np.float32)))
    brow = dev(rng.integers(-8, 8, (K, N)).astype(np.int8))
    s_t = dev(f16_bits(rng.integers(1, 3, N).astype(np.float32)))

    def prob(c_t):
        return capi.fpAintB_problem(a_t.data_ptr(), brow.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K, b_dtype=capi.I8,
                                    b_layout=capi.ROW_MAJOR, ldb=N)

    outs = []
    for which in range(2):
        c_t = torch.full((M, N), 0x7BFF, dtype=torch.int16, device="cuda")
        p = prob(c_t)
        assert capi.workspace_bytes(p) == K * N > capi.NOALLOC_SCRATCH_BYTES
        if which == 0:
            capi.launch(p, None, st)
        else:
            capi.launch_noalloc(p, st)
        torch.cuda.synchronize()
        outs.append(host_u16(c_t).copy())
    np.testing.assert_array_equal(outs[1], outs[0])
    # with the workspace the matrix-core route runs; on these integer inputs (exact sums) it gives the same bits too
    c_t = torch.full((M, N), 0x7BFF, dtype=torch.int16, device="cuda")
    p = prob(c_t)
    ws = torch.empty(capi.workspace_bytes(p), dtype=torch.uint8, device="cuda")
    capi.launch(p, ws.data_ptr(), st)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host_u16(c_t), outs[0])
    # need == 0: one full wave of 256-tiles, no workspace use of any kind
    M, N, K = 1024, 1024, 512
    a_t = dev(f16_bits(rng.uniform(-1, 1, (M, K))))
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    b_t = pack_on_device(q, K, N)
    s_t = dev(f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N)))
    outs = []
    for which in range(2):
        c_t = torch.full((M, N), 0x7BFF, dtype=torch.int16, device="cuda")
        p = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K, flags=capi.FLAG_TILE_256)
        assert capi.workspace_bytes(p) == 0
        (capi.launch_noalloc(p, st) if which else capi.launch(p, None, st))
        torch.cuda.synchronize()
        outs.append(host_u16(c_t).copy())
    np.testing.assert_array_equal(outs[1], outs[0])
    assert (outs[0] != 0x7BFF).all()


def test_noalloc_int8_planner_split_uses_the_scratch():
    """The planner's own K split of the int8 x int8 GEMM (int32 partials) through the built-in scratch: the small-M streaming
    kernel and the 128-tile kernel, same bits as with a caller workspace and as the single-slice launch without one."""
    rng = np.random.default_rng(99)
    st = torch.cuda.current_stream().cuda_stream
    for (M, N, K, kernel) in ((300, 1024, 8192, "wq_gemm_i8_mfma_128"), (8, 1024, 4096, "wq_gemm_i8_skinny")):
        a_t = dev(rng.integers(-128, 128, (M, K)).astype(np.int8))
        b_t = dev(rng.integers(-128, 128, (N, K)).astype(np.int8))
        bias_t = dev(rng.integers(-50000, 50001, N).astype(np.int32))
        outs = []
        for which in ("workspace", "noalloc", "none"):
            c_t = torch.full((M, N), 77, dtype=torch.int8, device="cuda")
            p = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), None, c_t.data_ptr(), M, N, K, a_dtype=capi.I8, b_dtype=capi.I8,
                                     c_dtype=capi.I8, epilogue=capi.EPI_ADD_RELU_MUL_CLAMP, requant=2e-4, d0=bias_t.data_ptr())
            need = capi.workspace_bytes(p)
            assert capi.kernel_name(p) == kernel and 0 < need <= capi.NOALLOC_SCRATCH_BYTES
            if which == "workspace":
                ws = torch.empty(need, dtype=torch.uint8, device="cuda")
                capi.launch(p, ws.data_ptr(), st)
            elif which == "noalloc":
                capi.launch_noalloc(p, st)
            else:
                capi.launch(p, None, st)
            torch.cuda.synchronize()
            outs.append(c_t.cpu().numpy().copy())
        np.testing.assert_array_equal(outs[1], outs[0])
        np.testing.assert_array_equal(outs[2], outs[0])
        assert np.unique(outs[0]).size > 20
