"""Operands above 2 GiB: the reference refuses them (include/ck/tensor_operation/gpu/grid/
gridwise_fpAintB_gemm_wmma.hpp:501-507); the fp16 / bf16 kernels here address with 64 bits and take them.

The checks are size-independent: a block of rows (or columns) that lies wholly above the 2 GiB mark must equal, bit for
bit, a separate launch of the same kernel over just that block, and a few of its rows the CPU checker.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from composable_kernel_amd import capi
from gpu_util import bits_to_f32, check_err, f16_bits, host_u16, pack_on_device

pytestmark = pytest.mark.gpu

TWO_GIB = 1 << 31


def _launch(p):
    assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
    assert capi.workspace_bytes(p) == 0
    capi.launch(p, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


@pytest.mark.parametrize("flag,kernel", [(capi.FLAG_TILE_256, "wq_gemm_mfma_256"), (capi.FLAG_TILE_128, "wq_gemm_mfma_128")])
def test_activations_and_output_above_2_gib(port, flag, kernel):
    M, N, K, tail = (1 << 18) + 512, 4096, 4096, 512
    assert (M - tail) * K * 2 >= TWO_GIB and (M - tail) * N * 2 >= TWO_GIB
    rng = np.random.default_rng(31)
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    s_t = torch.from_numpy(s.view(np.int16)).cuda()
    gen = torch.Generator(device="cuda").manual_seed(31)
    a_t = torch.empty((M, K), dtype=torch.float16, device="cuda").uniform_(-1, 1, generator=gen)
    c_t = torch.full((M, N), 0x7BFF, dtype=torch.int16, device="cuda")
    p = capi.fpAintB_problem(a_t.data_ptr(), bp.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K, flags=flag)
    assert capi.kernel_name(p) == kernel
    _launch(p)
    # the same kernel over the last rows alone, and over the first rows alone
    for r0 in (M - tail, 0):
        c_s = torch.full((tail, N), 0x7BFF, dtype=torch.int16, device="cuda")
        ps = capi.fpAintB_problem(a_t[r0:].data_ptr(), bp.data_ptr(), s_t.data_ptr(), c_s.data_ptr(), tail, N, K, flags=flag)
        assert capi.kernel_name(ps) == kernel
        _launch(ps)
        assert torch.equal(c_t[r0:r0 + tail], c_s)
    rows = slice(M - 40, M - 8)
    a_rows = np.ascontiguousarray(a_t[rows].cpu().numpy().view(np.uint16))
    exp = port.fpAintB_gemm_f16(a_rows, q, s, 32, N, K, K, K, N).reshape(32, N)
    check_err(bits_to_f32(host_u16(c_t[rows]), capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


@pytest.mark.parametrize("M,kernel", [(256, "wq_gemm_mfma_256"), (8, "wq_gemm_skinny")])
def test_weights_above_2_gib(port, M, kernel):
    N, K, tail = (1 << 19) + 256, 4096, 256
    assert (N - tail) * K >= TWO_GIB
    rng = np.random.default_rng(32)
    gen = torch.Generator(device="cuda").manual_seed(32)
    b_t = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda", generator=gen)
    s = f16_bits(rng.uniform(2.0 ** -10, 2.0 ** -6, N))
    s_t = torch.from_numpy(s.view(np.int16)).cuda()
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    a_t = torch.from_numpy(a.view(np.int16)).cuda()
    c_t = torch.full((M, N), 0x7BFF, dtype=torch.int16, device="cuda")
    p = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K, b_dtype=capi.I8)
    assert capi.kernel_name(p).startswith(kernel)
    _launch(p)
    got = host_u16(c_t[:, N - tail:])
    if M == 256:   # tile kernel: the last column tile alone runs the same code over the same K order
        c_s = torch.full((M, tail), 0x7BFF, dtype=torch.int16, device="cuda")
        ps = capi.fpAintB_problem(a_t.data_ptr(), b_t[N - tail:].data_ptr(), s_t[N - tail:].data_ptr(), c_s.data_ptr(),
                                  M, tail, K, b_dtype=capi.I8, flags=capi.FLAG_TILE_256)
        _launch(ps)
        np.testing.assert_array_equal(got, host_u16(c_s))
    rows = min(M, 16)
    q_tail = np.ascontiguousarray(b_t[N - tail:].cpu().numpy())
    exp = port.fpAintB_gemm_f16(np.ascontiguousarray(a[:rows]), q_tail, s[N - tail:], rows, tail, K, K, K, tail).reshape(rows, tail)
    check_err(bits_to_f32(got[:rows], capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


@pytest.mark.parametrize("flag,kb", [(capi.FLAG_TILE_256, 1), (capi.FLAG_TILE_128, 1), (capi.FLAG_TILE_128, 2), (capi.FLAG_GENERIC, 1)])
def test_full_d_matrix_indexed_beyond_int32(flag, kb):
    """A full (M, N) D operand whose flat index m * ldd0 + n passes 2^31: the epilogues (tile kernels, the general kernel
    and the split-K reduce) must index it with 64 bits.  Rows above the mark equal a separate launch over those rows
    alone with the D pointer moved to their first row; Add with an integer-valued D is exact, so the comparison is too."""
    M, N, K, ldd, tail = (1 << 18) + 512, 512, 128, 8192, 512
    assert (M - tail) * ldd >= (1 << 31)     # (beyond INT_MAX from the first row of the tail on)
    gen = torch.Generator(device="cuda").manual_seed(41)
    q = torch.randint(-8, 8, (N, K), dtype=torch.int8, device="cuda", generator=gen)
    bp = torch.empty((N, K // 2), dtype=torch.uint8, device="cuda")
    capi.pack_int4(q.data_ptr(), K, K, N, bp.data_ptr(), K // 2, torch.cuda.current_stream().cuda_stream)
    s_t = torch.ones((N,), dtype=torch.float16, device="cuda")
    a_t = torch.randint(-2, 3, (M, K), device="cuda", generator=gen).half()
    d_t = torch.zeros((M, ldd), dtype=torch.float16, device="cuda")
    d_t[:, :N] = torch.randint(-64, 65, (M, N), device="cuda", generator=gen).half()
    c_t = torch.full((M, N), 0x7BFF, dtype=torch.int16, device="cuda")

    def run(a, d, c, rows):
        p = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), s_t.data_ptr(), c.data_ptr(), rows, N, K, epilogue=capi.EPI_ADD,
                                 d0=d.data_ptr(), ldd0=ldd, flags=flag, k_batch=kb)
        assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
        ws = torch.empty(max(capi.workspace_bytes(p), 1), dtype=torch.uint8, device="cuda")
        capi.launch(p, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()

    run(a_t, d_t, c_t, M)
    r0 = M - tail
    c_s = torch.full((tail, N), 0x7BFF, dtype=torch.int16, device="cuda")
    run(a_t[r0:], d_t[r0:], c_s, tail)
    assert torch.equal(c_t[r0:], c_s)
    # and against plain arithmetic: every sum is a small integer, exact in fp16
    exp = (a_t[r0:].float() @ q.float().T + d_t[r0:, :N].float()).half()
    assert torch.equal(c_s.view(torch.float16), exp)
    # a leading dimension shorter than a row is refused
    bad = capi.fpAintB_problem(a_t.data_ptr(), bp.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), 256, N, K, epilogue=capi.EPI_ADD,
                               d0=d_t.data_ptr(), ldd0=N - 8)
    assert not capi.is_supported(bad)
