"""Parity at the configurations bench.py measures: the pre-shuffled weight orders (WQ_I4_PRESHUFFLED,
WQ_I8_PRESHUFFLED) at the full BASELINE sizes.  At these sizes the column-block and K-tile strides of the pre-shuffled
address map reach values no small case does, so every configuration is checked two ways: the whole output bit-equal to
the same launch on the plain weight order (same kernel, same arithmetic), and a row block against the CPU checker."""
import numpy as np
import pytest
import torch

from composable_kernel_amd import capi
from gpu_util import bits_to_f32, check_err

pytestmark = pytest.mark.gpu

MFMA256 = "wq_gemm_mfma_256"
F8_MFMA = "wq_gemm_f8_mfma_256"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _launch(p, ws=None):
    capi.launch(p, ws.data_ptr() if ws is not None else None, _stream())
    torch.cuda.synchronize()


def _pack(q, K, N):
    bp = torch.empty((N, K // 2), dtype=torch.uint8, device="cuda")
    capi.pack_int4(q.data_ptr(), K, K, N, bp.data_ptr(), K // 2, _stream())
    pre = torch.empty_like(bp)
    capi.preshuffle_int4(bp.data_ptr(), K // 2, K, N, pre.data_ptr(), K, _stream())
    torch.cuda.synchronize()
    return bp, pre


def _u16(t):
    return t.cpu().numpy().view(np.uint16)


def test_full_size_c2_preshuffled(port):
    """Config 2 (fp16 x int4, 4096^3) exactly as bench.py runs it."""
    M = N = K = 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(11939)
    a = (torch.rand((M, K), generator=g, device="cuda") * 2 - 1).half()
    q = torch.randint(-8, 8, (N, K), generator=g, device="cuda", dtype=torch.int8)
    s = (torch.rand((N,), generator=g, device="cuda") * (2.0 ** -4 - 2.0 ** -8) + 2.0 ** -8).half()
    bp, pre = _pack(q, K, N)
    c0 = torch.empty((M, N), dtype=torch.int16, device="cuda")
    c1 = torch.empty_like(c0)
    p0 = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), s.data_ptr(), c0.data_ptr(), M, N, K)
    p1 = capi.fpAintB_problem(a.data_ptr(), pre.data_ptr(), s.data_ptr(), c1.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)
    assert capi.kernel_name(p0) == MFMA256 and capi.kernel_name(p1) == MFMA256
    _launch(p0)
    _launch(p1)
    assert torch.equal(c0, c1)
    rows = slice(2500, 2516)
    exp = port.fpAintB_gemm_f16(_u16(a[rows].contiguous()), q.cpu().numpy(), _u16(s), 16, N, K, K, K, N).reshape(16, N)
    check_err(bits_to_f32(_u16(c1[rows]), capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
    # integer inputs: bit-exact against the checker
    ai = torch.randint(-1, 2, (M, K), generator=g, device="cuda").half()
    si = torch.randint(1, 3, (N,), generator=g, device="cuda").half()
    p1 = capi.fpAintB_problem(ai.data_ptr(), pre.data_ptr(), si.data_ptr(), c1.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)
    _launch(p1)
    rows = slice(4080, 4096)
    exp = port.fpAintB_gemm_f16(_u16(ai[rows].contiguous()), q.cpu().numpy(), _u16(si), 16, N, K, K, K, N).reshape(16, N)
    np.testing.assert_array_equal(_u16(c1[rows]), exp)


def test_full_size_c3_preshuffled(port):
    """Config 3 (bf16 x int8, 4096 x 11008 x 4096) on WQ_I8_PRESHUFFLED weights."""
    M, N, K = 4096, 11008, 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(33)
    a = (torch.rand((M, K), generator=g, device="cuda") * 2 - 1).bfloat16()
    s = (torch.rand((N,), generator=g, device="cuda") * (2.0 ** -6 - 2.0 ** -10) + 2.0 ** -10).bfloat16()
    b = torch.randint(-128, 128, (N, K), generator=g, device="cuda", dtype=torch.int8)
    pre = torch.empty_like(b)
    capi.preshuffle_int8(b.data_ptr(), K, K, N, pre.data_ptr(), K, _stream())
    c0 = torch.empty((M, N), dtype=torch.int16, device="cuda")
    c1 = torch.empty_like(c0)

    def both(a_, s_):
        p0 = capi.fpAintB_problem(a_.data_ptr(), b.data_ptr(), s_.data_ptr(), c0.data_ptr(), M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8)
        p1 = capi.fpAintB_problem(a_.data_ptr(), pre.data_ptr(), s_.data_ptr(), c1.data_ptr(), M, N, K, a_dtype=capi.BF16,
                                  b_dtype=capi.I8_PRESHUFFLED)
        assert capi.kernel_name(p0) == MFMA256 and capi.kernel_name(p1) == MFMA256
        _launch(p0)
        _launch(p1)
        assert torch.equal(c0, c1)

    both(a, s)
    rows = slice(1234, 1242)
    b_h = b.cpu().numpy()
    exp = port.gemm_bf16_i8(_u16(a[rows].contiguous().view(torch.int16)), b_h, _u16(s.view(torch.int16)), 8, N, K, K, 1, K, N, 0)
    check_err(bits_to_f32(_u16(c1[rows]), capi.BF16), bits_to_f32(exp.reshape(8, N), capi.BF16), 1e-2, 1e-2)
    ai = torch.randint(-1, 2, (M, K), generator=g, device="cuda").bfloat16()
    si = torch.randint(1, 3, (N,), generator=g, device="cuda").bfloat16()
    both(ai, si)
    rows = slice(4088, 4096)
    exp = port.gemm_bf16_i8(_u16(ai[rows].contiguous().view(torch.int16)), b_h, _u16(si.view(torch.int16)), 8, N, K, K, 1, K, N, 0)
    np.testing.assert_array_equal(_u16(c1[rows]), exp.reshape(8, N))


def test_full_size_c4_slices_preshuffled(port):
    """Config 4 per-GPU slices (fp16 x int4, M = N = 4096, K = 65536 in 8 slices of 8192) on pre-shuffled weights, in both
    forms a deployment can hold them: one pre-shuffled matrix over the whole K addressed by k_begin / k_end, and (as
    bench.py --gpus N stores them) one pre-shuffled matrix per slice.  Integer inputs: each fp32 partial equals the plain
    order's bit for bit, and their sum + finalize equals the single-pass kernel over the whole K."""
    M = N = 4096
    K, G = 65536, 8
    g = torch.Generator(device="cuda")
    g.manual_seed(44)
    q = torch.randint(-8, 8, (N, K), generator=g, device="cuda", dtype=torch.int8)
    bp, pre = _pack(q, K, N)
    ai = torch.randint(-1, 2, (M, K), generator=g, device="cuda").half()
    si = torch.randint(1, 3, (N,), generator=g, device="cuda").half()
    kread = capi.splitk_kread(K, G)
    total = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    part0 = torch.empty((M, N), dtype=torch.float32, device="cuda")
    part1 = torch.empty_like(part0)
    part2 = torch.empty_like(part0)
    for gi in range(G):
        k0, k1 = gi * kread, (gi + 1) * kread
        p0 = capi.fpAintB_problem(ai.data_ptr(), bp.data_ptr(), si.data_ptr(), part0.data_ptr(), M, N, K, k_begin=k0, k_end=k1,
                                  c_dtype=capi.F32)
        p1 = capi.fpAintB_problem(ai.data_ptr(), pre.data_ptr(), si.data_ptr(), part1.data_ptr(), M, N, K, k_begin=k0, k_end=k1,
                                  c_dtype=capi.F32, b_dtype=capi.I4_PRESHUFFLED)
        assert capi.kernel_name(p1) == MFMA256
        _launch(p0)
        _launch(p1)
        assert torch.equal(part0, part1), f"slice {gi}"
        if gi in (0, G - 1):
            # the slice as its own matrix: activations are a strided view, the weights of the slice packed and
            # pre-shuffled on their own
            _, pre_g = _pack(q[:, k0:k1].contiguous(), kread, N)
            a_g = ai[:, k0:k1]
            p2 = capi.fpAintB_problem(a_g.data_ptr(), pre_g.data_ptr(), si.data_ptr(), part2.data_ptr(), M, N, kread, lda=K,
                                      k_begin=0, k_end=kread, c_dtype=capi.F32, b_dtype=capi.I4_PRESHUFFLED)
            _launch(p2)
            assert torch.equal(part0, part2), f"slice {gi} as its own matrix"
            exp = port.fpAintB_partial(_u16(ai[:8].contiguous()), q.cpu().numpy(), _u16(si), 8, N, K, K, k0, k1)
            np.testing.assert_array_equal(part1[:8].cpu().numpy(), exp.reshape(8, N))
        total += part1
    c_sum = torch.empty((M, N), dtype=torch.int16, device="cuda")
    c_one = torch.empty((M, N), dtype=torch.int16, device="cuda")
    p = capi.fpAintB_problem(ai.data_ptr(), pre.data_ptr(), si.data_ptr(), c_sum.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)
    capi.reduce_finalize(p, total.data_ptr(), 1, _stream())
    p1 = capi.fpAintB_problem(ai.data_ptr(), pre.data_ptr(), si.data_ptr(), c_one.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)
    assert capi.kernel_name(p1) == MFMA256
    _launch(p1)
    assert torch.equal(c_sum, c_one)


@pytest.mark.parametrize("M", [4096, 3840])
def test_full_size_c5_preshuffled(port, M):
    """Config 5 (fp8 x int4 + MultiplyMultiply + FastGelu) on pre-shuffled weights: 4096^3 as bench.py runs it and
    example/65's default 3840 x 4096 x 4096."""
    N = K = 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(55 + M)
    a = (torch.randint(24, 88, (M, K), generator=g, device="cuda", dtype=torch.int16) +
         128 * torch.randint(0, 2, (M, K), generator=g, device="cuda", dtype=torch.int16)).to(torch.uint8)
    q = torch.randint(-8, 8, (N, K), generator=g, device="cuda", dtype=torch.int8)
    bp, pre = _pack(q, K, N)
    d0 = torch.rand((N,), generator=g, device="cuda") * 2.0 ** -4
    d1 = torch.rand((M,), generator=g, device="cuda")
    c0 = torch.empty((M, N), dtype=torch.int16, device="cuda")
    c1 = torch.empty_like(c0)
    kw = dict(a_dtype=capi.F8_FNUZ, c_dtype=capi.F16, flags=capi.FLAG_F8_NO_SPECIALS)
    for gelu, kind in ((False, capi.EPI_MULTIPLY_MULTIPLY), (True, capi.EPI_MULTIPLY_MULTIPLY_FASTGELU)):
        p0 = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), None, c0.data_ptr(), M, N, K, epilogue=kind, d0=d0.data_ptr(),
                                  d1=d1.data_ptr(), **kw)
        p1 = capi.fpAintB_problem(a.data_ptr(), pre.data_ptr(), None, c1.data_ptr(), M, N, K, epilogue=kind, d0=d0.data_ptr(),
                                  d1=d1.data_ptr(), b_dtype=capi.I4_PRESHUFFLED, **kw)
        assert capi.kernel_name(p0) == F8_MFMA and capi.kernel_name(p1) == F8_MFMA
        _launch(p0)
        _launch(p1)
        assert torch.equal(c0, c1)
        rows = slice(M - 24, M - 8)
        exp = port.gemm_f8_i8_mm(np.ascontiguousarray(a[rows].cpu().numpy()), q.cpu().numpy(), d0.cpu().numpy(),
                                 np.ascontiguousarray(d1[rows].cpu().numpy()), 16, N, K, K, K, N, gelu).reshape(16, N)
        check_err(bits_to_f32(_u16(c1[rows]), capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
    # exact set: activations in {-2 .. 2} (FNUZ bytes), power-of-two factors: bit-exact against the checker
    lut = torch.tensor([0xC8, 0xC0, 0x00, 0x40, 0x48], dtype=torch.uint8, device="cuda")
    ae = lut[torch.randint(0, 5, (M, K), generator=g, device="cuda")]
    d0e = torch.pow(2.0, torch.randint(-3, 1, (N,), generator=g, device="cuda").float())
    d1e = torch.pow(2.0, torch.randint(-3, 1, (M,), generator=g, device="cuda").float())
    p1 = capi.fpAintB_problem(ae.data_ptr(), pre.data_ptr(), None, c1.data_ptr(), M, N, K, epilogue=capi.EPI_MULTIPLY_MULTIPLY,
                              d0=d0e.data_ptr(), d1=d1e.data_ptr(), b_dtype=capi.I4_PRESHUFFLED, **kw)
    _launch(p1)
    rows = slice(0, 16)
    exp = port.gemm_f8_i8_mm(np.ascontiguousarray(ae[rows].cpu().numpy()), q.cpu().numpy(), d0e.cpu().numpy(),
                             np.ascontiguousarray(d1e[rows].cpu().numpy()), 16, N, K, K, K, N, False).reshape(16, N)
    np.testing.assert_array_equal(_u16(c1[rows]), exp)
