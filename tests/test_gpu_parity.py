"""GPU parity tests: the HIP kernels, driven through the C ABI, against the CPU checker
(oracle/wq_oracle.c) and the committed golden vectors produced by the reference's host code.

Bars: bit-exact for integer/index work (int4 packing, int8 requant GEMM, the generic kernel whose
fp32 accumulation runs in the reference's ascending-k order, and every kernel on integer-valued
inputs); for floating-point inputs on the MFMA kernels the accumulate order differs, so the bar is
ck::utils::check_err's defaults for the output type (half / bf16: rtol 1e-3, atol 1e-3 --
library/include/ck/library/utility/check_err.hpp:82-86,133-137) unless a test states otherwise.
"""
import numpy as np
import pytest
import torch

from composable_kernel_amd import capi
from gpu_util import (bf16_bits, bits_to_f32, check_err, dev, f16_bits, host_u16, pack_on_device,
                      run_fpaintb)

pytestmark = pytest.mark.gpu

MFMA = "wq_gemm_mfma_128"
MFMA256 = "wq_gemm_mfma_256"
GENERIC = "wq_gemm_generic"


def test_pack_int4_device_bit_exact(port):
    rng = np.random.default_rng(0)
    for K, N in ((64, 3), (4096, 257), (8, 1)):
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        got = pack_on_device(q, K, N).cpu().numpy()
        np.testing.assert_array_equal(got, port.pack_int4(q, K, N))


def preshuffle_on_device(packed: torch.Tensor, K: int, N: int, ldk: int | None = None) -> torch.Tensor:
    ldk = ldk or K
    out = torch.zeros(N * ldk // 2, dtype=torch.uint8, device="cuda")
    capi.preshuffle_int4(packed.data_ptr(), K // 2, K, N, out.data_ptr(), ldk, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


def test_preshuffle_int4_device_bit_exact(port):
    rng = np.random.default_rng(1)
    for K, N, ldk in ((64, 32, 64), (4096, 288, 4096), (128, 64, 192)):
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        packed = pack_on_device(q, K, N)
        got = preshuffle_on_device(packed, K, N, ldk).cpu().numpy()
        np.testing.assert_array_equal(got, port.preshuffle_int4(packed.cpu().numpy(), K, N, ldk))


@pytest.mark.parametrize("a_dtype", [capi.F16, capi.BF16])
def test_preshuffled_weights_equal_packed_weights(port, a_dtype):
    """Pre-shuffled int4 weights feed the same dequantisation and the same MFMA order as packed ones: every variant of
    the 256-tile kernel must return the packed run's bits on random (non-integer) data -- default and tuning
    schedules, both wave grids, partial last row tile, split-K, a K slice that starts inside the matrix, a padded K
    pitch, group-wise scales, the small-M case, and the general kernel reading the layout element by element; one case
    against the checker."""
    rng = np.random.default_rng(23)
    bits = f16_bits if a_dtype == capi.F16 else bf16_bits
    cases = [(256, 256, 256, 0, 1), (300, 512, 192, 0, 1), (512, 256, 512, capi.FLAG_LOCKSTEP, 1),
             (256, 512, 256, capi.FLAG_WAVES_2X4, 1), (256, 256, 384, capi.FLAG_DEQUANT_IN_L, 1),
             (256, 256, 256, capi.FLAG_WAVES_2X4 | capi.FLAG_DEQUANT_IN_L, 1), (256, 256, 512, 0, 2), (40, 256, 256, 0, 1),
             (256, 256, 128, capi.FLAG_DIRECT_STORE, 1)]
    for (M, N, K, flags, kb) in cases:
        a = bits(rng.uniform(-1, 1, (M, K)))
        s = bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        packed = pack_on_device(q, K, N)
        pre = preshuffle_on_device(packed, K, N)
        want = run_fpaintb(a, packed, s, M, N, K, a_dtype=a_dtype, flags=flags | capi.FLAG_TILE_256, k_batch=kb,
                           expect_kernel=MFMA256)
        got = run_fpaintb(a, pre, s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I4_PRESHUFFLED, flags=flags | capi.FLAG_TILE_256,
                          k_batch=kb, expect_kernel=MFMA256)
        np.testing.assert_array_equal(got, want)
    # against the checker, and the general kernel on the pre-shuffled bytes (bit-exact: ascending k)
    M, N, K = 64, 256, 128
    a = bits(rng.uniform(-1, 1, (M, K)))
    s = bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    pre = preshuffle_on_device(pack_on_device(q, K, N), K, N)
    gen = run_fpaintb(a, pre, s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I4_PRESHUFFLED, flags=capi.FLAG_GENERIC,
                      expect_kernel=GENERIC)
    if a_dtype == capi.F16:
        exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    else:
        exp = port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)
    np.testing.assert_array_equal(gen, exp)
    got = run_fpaintb(a, pre, s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I4_PRESHUFFLED, flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
    check_err(bits_to_f32(got, a_dtype), bits_to_f32(exp, a_dtype), 1e-2 if a_dtype == capi.BF16 else 1e-3, 1e-3)
    # padded K pitch (ldb = K + 64) and a K slice [128, 384) of K = 512 into an fp32 partial
    M, N, K, ldk = 256, 256, 512, 576
    a = bits(rng.uniform(-1, 1, (M, K)))
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    packed = pack_on_device(q, K, N)
    pre = preshuffle_on_device(packed, K, N, ldk)
    want = run_fpaintb(a, packed, s, M, N, K, a_dtype=a_dtype, flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
    got = run_fpaintb(a, pre, s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I4_PRESHUFFLED, ldb=ldk, flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
    np.testing.assert_array_equal(got, want)
    if a_dtype == capi.F16:
        # group-wise scales (groups of 128 k)
        sg = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, (K // 128, N)))
        want = run_fpaintb(a, packed, sg, M, N, K, flags=capi.FLAG_TILE_256, scale_group_k=128, expect_kernel=MFMA256)
        got = run_fpaintb(a, pre, sg, M, N, K, b_dtype=capi.I4_PRESHUFFLED, ldb=ldk, scale_group_k=128, flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("a_dtype", [capi.F16, capi.BF16])
def test_preshuffled_int8_weights_equal_plain_int8(port, a_dtype):
    """8-bit weights in their pre-shuffled order: the device re-layout against the checker's statement of the format,
    and every variant of the 256-tile kernel bit-equal to the run on plain int8 weights (random data); the general
    kernel on the pre-shuffled bytes against the checker."""
    rng = np.random.default_rng(37)
    bits = f16_bits if a_dtype == capi.F16 else bf16_bits

    def pre8(q, K, N, ldk=None):
        ldk = ldk or K
        src = dev(q)
        out = torch.zeros(N * ldk, dtype=torch.int8, device="cuda")
        capi.preshuffle_int8(src.data_ptr(), K, K, N, out.data_ptr(), ldk, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), port.preshuffle_int8(q, K, N, ldk))
        return out

    for (M, N, K, flags, kb, ldk) in ((256, 256, 256, 0, 1, None), (300, 512, 192, 0, 1, None), (512, 256, 512, capi.FLAG_LOCKSTEP, 1, None),
                                      (256, 512, 256, capi.FLAG_WAVES_2X4, 1, None), (256, 256, 384, capi.FLAG_DEQUANT_IN_L, 1, None),
                                      (256, 256, 512, 0, 2, None), (40, 256, 256, 0, 1, 320)):
        a = bits(rng.uniform(-1, 1, (M, K)))
        s = bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
        q = rng.integers(-128, 128, (N, K)).astype(np.int8)
        want = run_fpaintb(a, q, s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I8, flags=flags | capi.FLAG_TILE_256, k_batch=kb,
                           expect_kernel=MFMA256)
        got = run_fpaintb(a, pre8(q, K, N, ldk), s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I8_PRESHUFFLED, ldb=ldk,
                          flags=flags | capi.FLAG_TILE_256, k_batch=kb, expect_kernel=MFMA256)
        np.testing.assert_array_equal(got, want)
    M, N, K = 48, 256, 128
    a = bits(rng.uniform(-1, 1, (M, K)))
    s = bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    q = rng.integers(-128, 128, (N, K)).astype(np.int8)
    gen = run_fpaintb(a, pre8(q, K, N), s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I8_PRESHUFFLED, flags=capi.FLAG_GENERIC,
                      expect_kernel=GENERIC)
    if a_dtype == capi.F16:
        exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    else:
        exp = port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)
    np.testing.assert_array_equal(gen, exp)


@pytest.mark.parametrize("bits4", [True, False])
def test_preshuffled_weights_small_m_kernels(port, bits4):
    """The small-M kernels on pre-shuffled weights (packed int4 and int8): whatever kernel the planner picks for the
    plain layout it picks for the pre-shuffled one, and the output bits agree (same arithmetic and summation order; only
    the addresses of the weight stream differ).  Covers the grid-split kernel with 1, 2 and 4 row blocks and with split-K,
    and both four-waves-split-K kernels."""
    rng = np.random.default_rng(41 + bits4)
    for (M, N, K) in ((1, 2048, 4096), (8, 2048, 512), (16, 4096, 8192), (24, 128, 256), (40, 256, 1024), (64, 512, 4096),
                      (5, 4096, 16384), (33, 11008 // 64 * 64, 4096)):
        a = f16_bits(rng.uniform(-1, 1, (M, K)))
        s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
        if bits4:
            q = rng.integers(-8, 8, (N, K)).astype(np.int8)
            plain = pack_on_device(q, K, N)
            pre, plain_dt, pre_dt = preshuffle_on_device(plain, K, N), capi.I4_PACKED, capi.I4_PRESHUFFLED
        else:
            q = rng.integers(-128, 128, (N, K)).astype(np.int8)
            plain = dev(q)
            pre = torch.zeros(N * K, dtype=torch.int8, device="cuda")
            capi.preshuffle_int8(plain.data_ptr(), K, K, N, pre.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
            plain_dt, pre_dt = capi.I8, capi.I8_PRESHUFFLED
        pp = capi.fpAintB_problem(1 << 20, 1 << 20, 1 << 20, 1 << 20, M, N, K, b_dtype=plain_dt)
        name = capi.kernel_name(pp)
        assert name.startswith("wq_gemm_skinny")
        want = run_fpaintb(a, plain, s, M, N, K, b_dtype=plain_dt, expect_kernel=name)
        got = run_fpaintb(a, pre, s, M, N, K, b_dtype=pre_dt, expect_kernel=name)
        np.testing.assert_array_equal(got, want)
        if M * N * K <= 1 << 24:
            exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
            check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


@pytest.mark.parametrize("bits4", [True, False])
def test_preshuffled_weights_128_tile_kernel(port, bits4):
    """The 128-tile kernel on pre-shuffled weights: bit-equal to the plain layout (random data) for whole tiles, partial
    last row and column tiles (N a multiple of 32 only), bf16, planner-chosen split-K with a workspace, a K slice and
    group-wise scales."""
    rng = np.random.default_rng(43 + bits4)
    for (M, N, K, a_dtype, kb, gk) in ((128, 128, 128, capi.F16, 1, 0), (200, 160, 192, capi.F16, 1, 0), (256, 96, 256, capi.BF16, 1, 0),
                                       (512, 512, 2048, capi.F16, 1, 0), (128, 384, 512, capi.BF16, 2, 0), (130, 224, 256, capi.F16, 1, 64)):
        bits = f16_bits if a_dtype == capi.F16 else bf16_bits
        a = bits(rng.uniform(-1, 1, (M, K)))
        s = bits(rng.uniform(2.0 ** -8, 2.0 ** -4, (K // gk, N) if gk else N))
        if bits4:
            q = rng.integers(-8, 8, (N, K)).astype(np.int8)
            plain = pack_on_device(q, K, N)
            pre, plain_dt, pre_dt = preshuffle_on_device(plain, K, N), capi.I4_PACKED, capi.I4_PRESHUFFLED
        else:
            q = rng.integers(-128, 128, (N, K)).astype(np.int8)
            plain = dev(q)
            pre = torch.zeros(N * K, dtype=torch.int8, device="cuda")
            capi.preshuffle_int8(plain.data_ptr(), K, K, N, pre.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
            plain_dt, pre_dt = capi.I8, capi.I8_PRESHUFFLED
        kw = dict(a_dtype=a_dtype, flags=capi.FLAG_TILE_128, k_batch=kb, scale_group_k=gk, expect_kernel=MFMA)
        want = run_fpaintb(a, plain, s, M, N, K, b_dtype=plain_dt, **kw)
        got = run_fpaintb(a, pre, s, M, N, K, b_dtype=pre_dt, **kw)
        np.testing.assert_array_equal(got, want)
    # default routing of a mid-sized problem: the same kernel and planner split for both layouts
    M, N, K = 512, 1024, 4096
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    plain = pack_on_device(q, K, N)
    want = run_fpaintb(a, plain, s, M, N, K, expect_kernel=MFMA)
    got = run_fpaintb(a, preshuffle_on_device(plain, K, N), s, M, N, K, b_dtype=capi.I4_PRESHUFFLED, expect_kernel=MFMA)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("bits4", [True, False])
def test_group_wise_scales_small_m_kernels(port, bits4):
    """Group-wise weight scales in the small-M kernels (all three, with planner split-K, plain and pre-shuffled weights):
    integer-valued inputs bit-exact against the checker (pinned by the reference's fpAintB GEMM on the expanded scale
    tensor), decimal inputs within check_err's half tolerance, and a K slice that starts inside the matrix."""
    rng = np.random.default_rng(47 + bits4)
    qlo, qhi = (-8, 8) if bits4 else (-128, 128)
    plain_dt, pre_dt = (capi.I4_PACKED, capi.I4_PRESHUFFLED) if bits4 else (capi.I8, capi.I8_PRESHUFFLED)

    def layouts(q, K, N):
        if bits4:
            plain = pack_on_device(q, K, N)
            return plain, preshuffle_on_device(plain, K, N)
        plain = dev(q)
        pre = torch.zeros(N * K, dtype=torch.int8, device="cuda")
        capi.preshuffle_int8(plain.data_ptr(), K, K, N, pre.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
        return plain, pre

    for (M, N, K, g) in ((1, 2048, 4096, 128), (8, 2048, 512, 64), (16, 4096, 8192, 128), (24, 128, 256, 32),
                         (40, 256, 1024, 256), (64, 512, 2048, 128), (5, 4096, 16384, 128)):
        name = capi.kernel_name(capi.fpAintB_problem(1 << 20, 1 << 20, 1 << 20, 1 << 20, M, N, K, b_dtype=plain_dt, scale_group_k=g))
        assert name.startswith("wq_gemm_skinny"), name
        q = rng.integers(qlo, qhi, (N, K)).astype(np.int8)
        plain, pre = layouts(q, K, N)
        # exact set
        a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
        s = f16_bits(rng.integers(1, 4, (K // g, N)).astype(np.float32))
        exp = port.fpAintB_gemm_f16_grouped(a, q, s, M, N, K, K, K, N, g, N).reshape(M, N) if M * N * K <= 1 << 25 else None
        got = run_fpaintb(a, plain, s, M, N, K, b_dtype=plain_dt, scale_group_k=g, expect_kernel=name)
        got_pre = run_fpaintb(a, pre, s, M, N, K, b_dtype=pre_dt, scale_group_k=g, expect_kernel=name)
        np.testing.assert_array_equal(got_pre, got)
        if exp is not None:
            np.testing.assert_array_equal(got, exp)
        else:  # too long for the CPU checker: the 128-tile kernel (checked against it elsewhere) on the same exact inputs
            ref = run_fpaintb(a, plain, s, M, N, K, b_dtype=plain_dt, scale_group_k=g, flags=capi.FLAG_TILE_128, expect_kernel=MFMA)
            np.testing.assert_array_equal(got, ref)
        # decimal set
        if exp is not None:
            a = f16_bits(rng.uniform(-1, 1, (M, K)))
            s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, (K // g, N)))
            got = run_fpaintb(a, pre, s, M, N, K, b_dtype=pre_dt, scale_group_k=g, expect_kernel=name)
            exp = port.fpAintB_gemm_f16_grouped(a, q, s, M, N, K, K, K, N, g, N).reshape(M, N)
            check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


def test_preshuffled_weights_fp8_kernel(port):
    """The fp8 x int4 kernel on pre-shuffled weights: bit-equal to the packed run (same conversion, same MFMA order) on
    random bytes, both byte formats, a partial last row tile and split-K; exact set against the checker."""
    rng = np.random.default_rng(29)
    for (M, N, K, kb) in ((256, 256, 128, 1), (300, 512, 512, 1), (512, 256, 1024, 2), (100, 256, 384, 1)):
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        packed = pack_on_device(q, K, N)
        pre = preshuffle_on_device(packed, K, N)
        d0 = rng.uniform(0.0, 1.0, N).astype(np.float32)
        d1 = rng.uniform(0.0, 1.0, M).astype(np.float32)
        a = _f8_bytes(rng, (M, K))
        for adt, fl in ((capi.F8_FNUZ, capi.FLAG_F8_NO_SPECIALS), (capi.F8_E4M3FN, 0)):
            kw = dict(a_dtype=adt, epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU, d0=d0, d1=d1, flags=fl, k_batch=kb,
                      expect_kernel=F8_MFMA)
            want = run_fpaintb(a, packed, None, M, N, K, **kw)
            got = run_fpaintb(a, pre, None, M, N, K, b_dtype=capi.I4_PRESHUFFLED, **kw)
            np.testing.assert_array_equal(got, want)
    M, N, K = 256, 256, 256
    lut = {0: 0x00, 1: 0x40, 2: 0x48, -1: 0xC0, -2: 0xC8}
    a = np.vectorize(lut.get)(rng.integers(-2, 3, (M, K))).astype(np.uint8)
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    pre = preshuffle_on_device(pack_on_device(q, K, N), K, N)
    d0e = (2.0 ** rng.integers(-3, 1, N)).astype(np.float32)
    d1e = (2.0 ** rng.integers(-3, 1, M)).astype(np.float32)
    got = run_fpaintb(a, pre, None, M, N, K, a_dtype=capi.F8_FNUZ, b_dtype=capi.I4_PRESHUFFLED, epilogue=capi.EPI_MULTIPLY_MULTIPLY,
                      d0=d0e, d1=d1e, flags=capi.FLAG_F8_NO_SPECIALS, expect_kernel=F8_MFMA)
    np.testing.assert_array_equal(got, port.gemm_f8_i8_mm(a, q, d0e, d1e, M, N, K, K, K, N, False).reshape(M, N))
    # the four-section form reads weights through LDS in the packed order only
    p = capi.fpAintB_problem(1 << 20, 1 << 20, None, 1 << 20, M, N, K, a_dtype=capi.F8_E4M3FN, b_dtype=capi.I4_PRESHUFFLED,
                             c_dtype=capi.F16, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=1 << 20, d1=1 << 20, flags=capi.FLAG_PINGPONG)
    assert not capi.is_supported(p)


def test_mfma_layout_probe(port):
    """A = shifted identity pattern, B asymmetric: catches transposed / permuted fragment maps."""
    M = N = 128
    K = 64
    a = np.zeros((M, K), np.float16)
    for m in range(M):
        a[m, (m * 7 + 3) % K] = 1.0        # one 1 per row at a row-dependent k
    q = np.zeros((N, K), np.int8)
    for n in range(N):
        for k in range(K):
            q[n, k] = ((n * 5 + k * 3) % 16) - 8   # asymmetric in (n, k)
    s = f16_bits(np.ones(N))
    got = run_fpaintb(f16_bits(a), pack_on_device(q, K, N), s, M, N, K, expect_kernel=MFMA)
    exp = port.fpAintB_gemm_f16(f16_bits(a), q, s, M, N, K, K, K, N).reshape(M, N)
    np.testing.assert_array_equal(got, exp)


def test_golden_integer_case_exact_on_mfma(golden_gemm):
    g = golden_gemm
    M, N, K, lda, ldb, ldc = (int(v) for v in g["fpaintb_int_shape"])
    q = np.ascontiguousarray(g["fpaintb_int_b"][:, :K])
    got = run_fpaintb(g["fpaintb_int_a"], pack_on_device(q, K, N), g["fpaintb_int_s"], M, N, K,
                      lda=lda, ldc=ldc, expect_kernel=MFMA)
    np.testing.assert_array_equal(got.reshape(-1)[: g["fpaintb_int_c"].size], g["fpaintb_int_c"])
    assert ldc == N


@pytest.mark.parametrize("tag", ["fpaintb_real", "fpaintb_ragged"])
def test_golden_cases_generic_kernel_bit_exact(golden_gemm, tag):
    """The generic kernel accumulates in ascending k with separate multiply/add: bit-exact."""
    g = golden_gemm
    M, N, K, lda, ldb, ldc = (int(v) for v in g[f"{tag}_shape"])
    q = np.ascontiguousarray(g[f"{tag}_b"][:, :K])
    got = run_fpaintb(g[f"{tag}_a"], pack_on_device(q, K, N), g[f"{tag}_s"], M, N, K, lda=lda, ldc=ldc,
                      flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
    exp = np.zeros(M * ldc, np.uint16)
    exp[: g[f"{tag}_c"].size] = g[f"{tag}_c"]
    # compare only the M x N window (the padding columns of ldc keep their poison)
    np.testing.assert_array_equal(got[:, :N], exp.reshape(M, ldc)[:, :N])
    assert (got[:, N:] == 0x7BFF).all()


def test_golden_i8_weights_generic(golden_gemm):
    g = golden_gemm
    M, N, K, lda, ldb, ldc = (int(v) for v in g["fpaintb_i8_shape"])
    b = g["fpaintb_i8_b"]
    exp = g["fpaintb_i8_c"].reshape(M, N)
    got = run_fpaintb(g["fpaintb_i8_a"], b, g["fpaintb_i8_s"], M, N, K, b_dtype=capi.I8, flags=capi.FLAG_GENERIC,
                      expect_kernel=GENERIC)
    np.testing.assert_array_equal(got, exp)
    bu = (b.astype(np.int16) + 128).astype(np.uint8)   # example/64's UnsignedWeightPreprocessor
    got = run_fpaintb(g["fpaintb_i8_a"], bu, g["fpaintb_i8_s"], M, N, K, b_dtype=capi.U8_OFFSET,
                      flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (128, 256, 1024)])
def test_fp16_int4_mfma_vs_checker(port, M, N, K):
    rng = np.random.default_rng(M + N + K)
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    got = run_fpaintb(a, pack_on_device(q, K, N), s, M, N, K, expect_kernel=MFMA)
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
    # integer-valued inputs: every partial sum is exact in fp32, so any order gives the same bits
    a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = f16_bits(rng.integers(1, 4, N).astype(np.float32))
    got = run_fpaintb(a, pack_on_device(q, K, N), s, M, N, K, expect_kernel=MFMA)
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("b_dtype", [capi.I8, capi.U8_OFFSET])
def test_fp16_int8_mfma(port, b_dtype):
    rng = np.random.default_rng(11)
    M, N, K = 128, 256, 256
    q = rng.integers(-128, 128, (N, K)).astype(np.int8)
    b = q if b_dtype == capi.I8 else (q.astype(np.int16) + 128).astype(np.uint8)
    a = f16_bits(np.round(rng.uniform(-1, 1, (M, K))))
    s = f16_bits(rng.integers(1, 3, N).astype(np.float32))
    got = run_fpaintb(a, b, s, M, N, K, b_dtype=b_dtype, expect_kernel=MFMA)
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    np.testing.assert_array_equal(got, exp)
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -10, 2.0 ** -7, N))
    got = run_fpaintb(a, b, s, M, N, K, b_dtype=b_dtype, expect_kernel=MFMA)
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


def test_bf16_int8_golden_generic_all_epilogues(golden_gemm):
    """Config C3 semantics on the golden case (48 x 64 x 128 -> generic kernel)."""
    g = golden_gemm
    M, N, K = 48, 64, 128
    for ep, kind in ((0, capi.EPI_PASSTHROUGH), (1, capi.EPI_FASTGELU), (2, capi.EPI_MULTIPLY_ADD_FASTGELU)):
        got = run_fpaintb(g["bf16_a"], g["bf16_bcol"], g["bf16_s"], M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                          epilogue=kind, d0=g["bf16_d0"], d1=g["bf16_d1"], flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
        # (the default route for 48 rows is the small-M kernel: same result up to the summation order)
        dflt = run_fpaintb(g["bf16_a"], g["bf16_bcol"], g["bf16_s"], M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                           epilogue=kind, d0=g["bf16_d0"], d1=g["bf16_d1"], expect_kernel="wq_gemm_skinny")
        check_err(bits_to_f32(dflt, capi.BF16), bits_to_f32(got, capi.BF16), 1e-2, 1e-3)
        exp = g[f"bf16_e_col_ep{ep}"].reshape(M, N)
        if ep == 0:
            np.testing.assert_array_equal(got, exp)   # same order, truncating store: bit-exact
        else:
            # FastGelu on the device is the reference's device form (hardware exp2 and rcp,
            # unary_element_wise_operation.hpp:455-465) vs double exp and a true division on the host: one bf16
            # ulp, or 1e-6 absolute where the result underflows towards zero (x << 0)
            d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
            tiny = np.abs(bits_to_f32(got, capi.BF16) - bits_to_f32(exp, capi.BF16)) <= 1e-6
            assert (d[~tiny] <= 1).all(), d[~tiny].max()
    # row-major B (example/60's layout) through the generic kernel
    brow = np.ascontiguousarray(g["bf16_bcol"].T)
    got = run_fpaintb(g["bf16_a"], brow, g["bf16_s"], M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                      epilogue=capi.EPI_FASTGELU, b_layout=capi.ROW_MAJOR, ldb=N, expect_kernel=GENERIC)
    exp = g["bf16_e_row_ep1"].reshape(M, N)
    d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    tiny = np.abs(bits_to_f32(got, capi.BF16) - bits_to_f32(exp, capi.BF16)) <= 1e-6
    assert (d[~tiny] <= 1).all()


def test_bf16_int8_multi_abd_family_golden(golden_gemm):
    """The binary epilogues of the multi-ABD instance family on the golden case: Multiply on B with Add /
    AddFastGelu / Multiply / MultiplyFastGelu, and the "multiply" instances (no scale on B, row-major B)."""
    g = golden_gemm
    M, N, K = 48, 64, 128
    a, s, d0, d1 = g["bf16_a"], g["bf16_s"], g["bf16_d0"], g["bf16_d1"]
    kinds = {3: capi.EPI_ADD, 4: capi.EPI_ADD_FASTGELU, 5: capi.EPI_MULTIPLY, 6: capi.EPI_MULTIPLY_FASTGELU,
             0: capi.EPI_PASSTHROUGH, 2: capi.EPI_MULTIPLY_ADD_FASTGELU}

    def same(got, exp, gelu):
        exp = exp.reshape(M, N)
        if not gelu:
            np.testing.assert_array_equal(got, exp)
            return
        d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
        tiny = np.abs(bits_to_f32(got, capi.BF16) - bits_to_f32(exp, capi.BF16)) <= 1e-6
        assert (d[~tiny] <= 1).all()

    for ep in (3, 4, 5, 6):
        got = run_fpaintb(a, g["bf16_bcol"], s, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=kinds[ep], d0=d0,
                          flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
        same(got, g[f"bf16_e_col_ep{ep}"], ep in (4, 6))
    brow = np.ascontiguousarray(g["bf16_bcol"].T)
    for ep in (0, 2, 5, 6):
        got = run_fpaintb(a, brow, None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=kinds[ep], d0=s, d1=d1,
                          b_layout=capi.ROW_MAJOR, ldb=N, expect_kernel=GENERIC)
        same(got, g[f"bf16_mul_e_row_ep{ep}"], ep in (2, 6))


def test_bf16_int8_multi_abd_family_mfma(port):
    """Same family on the matrix-core kernels (tile-aligned, K-contiguous B), against the CPU checker."""
    rng = np.random.default_rng(30)
    M, N, K = 256, 256, 192
    q = rng.integers(-128, 128, (N, K)).astype(np.int8)
    a = bf16_bits(rng.uniform(-1, 1, (M, K)))
    s = bf16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    d1 = bf16_bits(rng.uniform(-0.5, 0.5, N))
    for flags, kern in ((0, MFMA), (capi.FLAG_TILE_256, MFMA256)):
        for ep, kind, with_scale in ((3, capi.EPI_ADD, True), (4, capi.EPI_ADD_FASTGELU, True), (5, capi.EPI_MULTIPLY, False),
                                     (6, capi.EPI_MULTIPLY_FASTGELU, False), (2, capi.EPI_MULTIPLY_ADD_FASTGELU, False)):
            d0 = d1 if with_scale else s
            got = run_fpaintb(a, q, s if with_scale else None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=kind,
                              d0=d0, d1=d1, flags=flags, expect_kernel=kern)
            exp = port.gemm_bf16_i8(a, q, s if with_scale else None, M, N, K, K, 1, K, N, ep, d0, d1).reshape(M, N)
            check_err(bits_to_f32(got, capi.BF16), bits_to_f32(exp, capi.BF16), 1e-2, 1e-3)


def test_bf16_multiply_add_bit_pattern_epilogue(golden_gemm, port):
    """MultiplyAdd<bhalf_t, float, bhalf_t, bhalf_t> (element_wise_operation.hpp:200-231): integer arithmetic on the
    16-bit patterns.  Goldens from the reference functor bit for bit on the general kernel (per-column and full-matrix
    D operands); the tile, small-M and split-K paths on integer-valued inputs, where the accumulator -- and with it the
    bit pattern the functor multiplies -- does not depend on the summation order."""
    g = golden_gemm
    M, N, K = 48, 64, 128
    brow = np.ascontiguousarray(g["bf16_bcol"].T)
    got = run_fpaintb(g["bf16_a"], brow, None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=capi.EPI_MULTIPLY_ADD,
                      d0=g["bf16_s"], d1=g["bf16_d1"], b_layout=capi.ROW_MAJOR, ldb=N, expect_kernel=GENERIC)
    np.testing.assert_array_equal(got, g["bf16_mul_e_row_ep7"].reshape(M, N))
    got = run_fpaintb(g["bf16_a"], brow, None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=capi.EPI_MULTIPLY_ADD,
                      d0=g["bf16_dm0"], d1=g["bf16_dm1"], ldd0=N, ldd1=N + 8, b_layout=capi.ROW_MAJOR, ldb=N,
                      expect_kernel=GENERIC)
    np.testing.assert_array_equal(got, g["bf16_dmat_e_row_ep7"].reshape(M, N))
    rng = np.random.default_rng(13)
    for (M, N, K, flags, kern, kb) in ((256, 256, 192, capi.FLAG_TILE_128, MFMA, 1), (300, 256, 192, capi.FLAG_TILE_256, MFMA256, 1),
                                       (256, 256, 256, capi.FLAG_TILE_256, MFMA256, 2), (24, 128, 256, 0, "wq_gemm_skinny", 1),
                                       (8, 2048, 512, 0, "wq_gemm_skinny_ws", 1)):
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        a = bf16_bits(np.round(rng.uniform(-2, 2, (M, K))))
        d0 = rng.integers(0, 65536, N).astype(np.uint16)          # any bit patterns: the functor treats them as integers
        d1 = rng.integers(0, 65536, N).astype(np.uint16)
        got = run_fpaintb(a, q, None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=capi.EPI_MULTIPLY_ADD, d0=d0, d1=d1,
                          flags=flags, k_batch=kb, expect_kernel=kern)
        exp = port.gemm_bf16_i8(a, q, None, M, N, K, K, 1, K, N, 7, d0, d1).reshape(M, N)
        np.testing.assert_array_equal(got, exp)
    # fp16 outputs have no such functor in the engine
    p = capi.fpAintB_problem(1 << 20, 1 << 20, None, 1 << 20, 256, 256, 256, a_dtype=capi.F16, b_dtype=capi.I8,
                             epilogue=capi.EPI_MULTIPLY_ADD, d0=1 << 20, d1=1 << 20)
    assert not capi.is_supported(p)


def test_bf16_convert_rtn_epilogue(golden_gemm, port):
    """ConvertBF16RTN (unary_element_wise_operation.hpp:236-251 -> type_convert.hpp:507-549) as the C operation: the
    accumulator leaves rounded to nearest even.  Goldens from the reference functor bit for bit on the general kernel;
    the tile, small-M and split-K paths on half-integer inputs, whose sums are exact in fp32 in any order and hit ties
    (differences between the engine's two conversions are asserted to occur)."""
    g = golden_gemm
    M, N, K = 48, 64, 128
    brow = np.ascontiguousarray(g["bf16_bcol"].T)
    got = run_fpaintb(g["bf16_a"], g["bf16_bcol"], g["bf16_s"], M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                      epilogue=capi.EPI_CONVERT_BF16_RTN, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
    np.testing.assert_array_equal(got, g["bf16_e_col_ep8"].reshape(M, N))
    for a_key, e_key in (("bf16_a", "bf16_mul_e_row_ep8"), ("rtn_a_tie", "bf16_tie_e_row_ep8")):
        got = run_fpaintb(g[a_key], brow, None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=capi.EPI_CONVERT_BF16_RTN,
                          b_layout=capi.ROW_MAJOR, ldb=N, expect_kernel=GENERIC)
        np.testing.assert_array_equal(got, g[e_key].reshape(M, N))
    rng = np.random.default_rng(17)
    differs = 0
    for (M, N, K, flags, kern, kb) in ((256, 256, 192, capi.FLAG_TILE_128, MFMA, 1), (300, 256, 192, capi.FLAG_TILE_256, MFMA256, 1),
                                       (256, 256, 256, capi.FLAG_TILE_256, MFMA256, 2), (24, 128, 256, 0, "wq_gemm_skinny", 1),
                                       (8, 2048, 512, 0, "wq_gemm_skinny_ws", 1)):
        q = rng.integers(-128, 128, (N, K)).astype(np.int8)
        a = bf16_bits(np.round(rng.uniform(-4, 4, (M, K)) * 2) / 2)   # multiples of 1/2: exact sums, nine-bit results and more
        got = run_fpaintb(a, q, None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=capi.EPI_CONVERT_BF16_RTN,
                          flags=flags, k_batch=kb, expect_kernel=kern)
        exp = port.gemm_bf16_i8(a, q, None, M, N, K, K, 1, K, N, 8).reshape(M, N)
        np.testing.assert_array_equal(got, exp)
        trunc = port.gemm_bf16_i8(a, q, None, M, N, K, K, 1, K, N, 0).reshape(M, N)
        differs += int((exp != trunc).sum())
    assert differs > 1000
    # realistic values through the 256-tile kernel with Multiply on B: within one bf16 step of the checker's nearest-even
    # result (fp32 summation order differs), never further
    M, N, K = 256, 512, 1024
    a = bf16_bits(rng.uniform(-1, 1, (M, K)))
    s = bf16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    q = rng.integers(-128, 128, (N, K)).astype(np.int8)
    got = run_fpaintb(a, q, s, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=capi.EPI_CONVERT_BF16_RTN,
                      flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
    exp = port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 8).reshape(M, N)
    check_err(bits_to_f32(got, capi.BF16), bits_to_f32(exp, capi.BF16), 1e-2, 1e-3)
    # fp16 outputs have no such conversion
    p = capi.fpAintB_problem(1 << 20, 1 << 20, None, 1 << 20, 256, 256, 256, a_dtype=capi.F16, b_dtype=capi.I8,
                             epilogue=capi.EPI_CONVERT_BF16_RTN)
    assert not capi.is_supported(p)


@pytest.mark.parametrize("a_dtype", [capi.F16, capi.BF16])
def test_row_major_b_via_transpose(port, a_dtype):
    """8-bit weights stored (K, N) row-major: with a workspace the launch transposes them and runs the matrix-core
    kernel (bit-identical to the K-contiguous call); without one the general kernel gives the checker's result."""
    rng = np.random.default_rng(60 + a_dtype)
    M, N, K = 256, 384, 320
    q = rng.integers(-128, 128, (N, K)).astype(np.int8)      # (N, K): K contiguous
    brow = np.ascontiguousarray(q.T)                          # (K, N): N contiguous
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits
    a = cvt(rng.uniform(-1, 1, (M, K)))
    s = cvt(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    col = run_fpaintb(a, q, s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I8, expect_kernel=MFMA)
    row = run_fpaintb(a, brow, s, M, N, K, a_dtype=a_dtype, b_dtype=capi.I8, b_layout=capi.ROW_MAJOR, ldb=N,
                      expect_kernel="wq_transpose_b8+" + MFMA)
    np.testing.assert_array_equal(row, col)
    # padded leading dimension and the offset-binary format
    pad = np.zeros((K, N + 16), np.uint8)
    pad[:, :N] = (brow.astype(np.int16) + 128).astype(np.uint8)
    row = run_fpaintb(a, pad, s, M, N, K, a_dtype=a_dtype, b_dtype=capi.U8_OFFSET, b_layout=capi.ROW_MAJOR, ldb=N + 16,
                      expect_kernel="wq_transpose_b8+" + MFMA)
    np.testing.assert_array_equal(row, col)


def test_bf16_int8_mfma(port):
    rng = np.random.default_rng(5)
    M, N, K = 256, 256, 512
    q = rng.integers(-128, 128, (N, K)).astype(np.int8)
    # exact set
    a = bf16_bits(np.round(rng.uniform(-1, 1, (M, K))))
    s = bf16_bits(rng.integers(1, 3, N).astype(np.float32))
    got = run_fpaintb(a, q, s, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, expect_kernel=MFMA)
    exp = port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)
    np.testing.assert_array_equal(got, exp)
    # realistic set: bf16 output is a truncation, so a different fp32 summation order can move the
    # result by one bf16 ulp (2^-8 relative); bar = 1 ulp on every element + check_err 1e-2.
    a = bf16_bits(rng.uniform(-1, 1, (M, K)))
    s = bf16_bits(rng.uniform(2.0 ** -10, 2.0 ** -7, N))
    got = run_fpaintb(a, q, s, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, expect_kernel=MFMA)
    exp = port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)
    check_err(bits_to_f32(got, capi.BF16), bits_to_f32(exp, capi.BF16), 1e-2, 1e-3)
    # bf16 x packed int4 (exact set)
    q4 = rng.integers(-8, 8, (N, K)).astype(np.int8)
    a = bf16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = bf16_bits(rng.integers(1, 3, N).astype(np.float32))
    got = run_fpaintb(a, pack_on_device(q4, K, N), s, M, N, K, a_dtype=capi.BF16, expect_kernel=MFMA)
    exp = port.gemm_bf16_i8(a, q4, s, M, N, K, K, 1, K, N, 0).reshape(M, N)
    np.testing.assert_array_equal(got, exp)


def _run_i8(a, b, M, N, K, requant, flags=0, lda=None, ldb=None, ldc=None, bias=None, relu=False, use_workspace=True):
    a_t, b_t = dev(a), dev(b)
    ldc = ldc or N
    e_t = torch.full((M + 16, ldc), 77, dtype=torch.int8, device="cuda")  # 16 guard rows
    bias_t = dev(np.ascontiguousarray(bias, np.int32)) if bias is not None else None
    epi = ((capi.EPI_ADD_RELU_MUL_CLAMP if relu else capi.EPI_ADD_ACT_MUL_CLAMP) if bias is not None
           else (capi.EPI_RELU_MUL_CLAMP if relu else capi.EPI_ACT_MUL_CLAMP))
    p = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), None, e_t.data_ptr(), M, N, K, a_dtype=capi.I8,
                             b_dtype=capi.I8, c_dtype=capi.I8, epilogue=epi, requant=requant,
                             d0=bias_t.data_ptr() if bias_t is not None else None,
                             flags=flags, lda=lda, ldb=ldb, ldc=ldc)
    assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
    nbytes = capi.workspace_bytes(p)   # the planner's own split-K of a short grid (optional)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda") if nbytes and use_workspace else None
    capi.launch(p, ws.data_ptr() if ws is not None else None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    e = e_t.cpu().numpy()
    assert (e[M:] == 77).all(), "rows past M were written"
    return e[:M], capi.kernel_name(p)


def test_c1_int8_requant_gemm_exact(golden_gemm, port):
    """example/14 problem (M=N=K=256, srand(1) inputs, requant 0.03) on the GPU: atol 0, on both the
    int8 MFMA kernel and the generic kernel."""
    M = N = K = 256
    a = port.generator2_i8(M, K, K, 1, -128, 127, seed=1)
    b = port.generator2_i8(K, N, 1, K, -128, 127, seed=-1)
    e, name = _run_i8(a, b, M, N, K, 0.03)
    assert name == "wq_gemm_i8_mfma_128"
    np.testing.assert_array_equal(e, golden_gemm["c1_e"])
    e, name = _run_i8(a, b, M, N, K, 0.03, flags=capi.FLAG_GENERIC)
    assert name == GENERIC
    np.testing.assert_array_equal(e, golden_gemm["c1_e"])


def test_int8_requant_golden_ragged_and_large(golden_gemm, port):
    g = golden_gemm
    # ragged strides / sizes (40 x 24 x 96): generic kernel, golden from the reference
    e, name = _run_i8(g["i8_small_a"], g["i8_small_b"], 40, 24, 96, 0.25, lda=104, ldb=100, ldc=32)
    assert name == GENERIC
    exp = np.full(40 * 32, 77, np.int8)
    exp[: g["i8_small_e"].size] = g["i8_small_e"]
    np.testing.assert_array_equal(e[:, :24], exp.reshape(40, 32)[:, :24])
    # example/14's own size (1024^3, |acc| <= 2^24 so the reference's float accumulate is exact): MFMA kernel
    # against the checker on a row block and against the generic kernel everywhere
    rng = np.random.default_rng(14)
    M = N = K = 1024
    a = rng.integers(-128, 128, (M, K)).astype(np.int8)
    b = rng.integers(-128, 128, (N, K)).astype(np.int8)
    e, name = _run_i8(a, b, M, N, K, 0.03)
    assert name == "wq_gemm_i8_mfma_128"
    eg, _ = _run_i8(a, b, M, N, K, 0.03, flags=capi.FLAG_GENERIC)
    np.testing.assert_array_equal(e, eg)
    ref = port.gemm_i8_requant(np.ascontiguousarray(a[:64]), b, 64, N, K, K, K, N, 0.03).reshape(64, N)
    np.testing.assert_array_equal(e[:64], ref)
    # small requant scale so that outputs are not saturated
    e, _ = _run_i8(a, b, M, N, K, 1e-4)
    ref = port.gemm_i8_requant(np.ascontiguousarray(a[:64]), b, 64, N, K, K, K, N, 1e-4).reshape(64, N)
    np.testing.assert_array_equal(e[:64], ref)
    assert np.abs(ref.astype(np.int32)).max() < 127 and np.unique(ref).size > 20


def test_int8_bias_relu_requant(golden_gemm, port):
    """Activation_Mul_Clamp<Relu> and Add_Activation_Mul_Clamp<PassThrough | Relu> (example/14's bias + relu
    variant): golden cases bit-exact on the int8 MFMA kernel and the generic kernel, the example's own 1024^3 size
    against the checker on a row block."""
    g = golden_gemm
    M = N = K = 128
    a = port.generator2_i8(M, K, K, 1, -128, 127, seed=1)
    b = port.generator2_i8(K, N, 1, K, -128, 127, seed=-1)
    for relu in (0, 1):
        for bias, key in ((None, f"i8b_e_relu{relu}"), (g["i8b_bias"], f"i8b_e_bias_relu{relu}")):
            e, name = _run_i8(a, b, M, N, K, 0.03, bias=bias, relu=bool(relu))
            assert name == "wq_gemm_i8_mfma_128"
            np.testing.assert_array_equal(e.ravel(), g[key])
            e, name = _run_i8(a, b, M, N, K, 0.03, bias=bias, relu=bool(relu), flags=capi.FLAG_GENERIC)
            assert name == GENERIC
            np.testing.assert_array_equal(e.ravel(), g[key])
        for bias, key in ((None, f"i8_small_e_relu{relu}"), (g["i8_small_bias"], f"i8_small_e_bias_relu{relu}")):
            e, name = _run_i8(g["i8_small_a"], g["i8_small_b"], 40, 24, 96, 0.25, lda=104, ldb=100, ldc=32, bias=bias,
                              relu=bool(relu))
            assert name == GENERIC
            exp = np.concatenate([g[key], np.zeros(40 * 32 - g[key].size, np.int8)])
            np.testing.assert_array_equal(e[:, :24], exp.reshape(40, 32)[:, :24])
    rng = np.random.default_rng(15)
    M = N = K = 1024
    a = rng.integers(-128, 128, (M, K)).astype(np.int8)
    b = rng.integers(-128, 128, (N, K)).astype(np.int8)
    bias = rng.integers(-300000, 300001, N).astype(np.int32)
    for rq in (0.03, 2e-4):
        e, name = _run_i8(a, b, M, N, K, rq, bias=bias, relu=True)
        assert name == "wq_gemm_i8_mfma_128"
        ref = port.gemm_i8_bias_requant(np.ascontiguousarray(a[:64]), b, bias, 64, N, K, K, K, N, rq, 1).reshape(64, N)
        np.testing.assert_array_equal(e[:64], ref)
    assert ref.min() == 0 and np.unique(ref).size > 20
    # ragged M stays on the matrix core: partial last row tile, guard rows after E untouched
    for Mr in (1, 100, 333):
        ar = np.ascontiguousarray(a[:Mr])
        e, name = _run_i8(ar, b, Mr, N, K, 2e-4, bias=bias, relu=True, flags=capi.FLAG_TILE_128)
        assert name == "wq_gemm_i8_mfma_128"
        ref = port.gemm_i8_bias_requant(ar, b, bias, Mr, N, K, K, K, N, 2e-4, 1).reshape(Mr, N)
        np.testing.assert_array_equal(e, ref)
    # the 256-tile kernel: forced at a small ragged size (all four functors), chosen by itself at 4096^3, and there
    # bit-equal to the 128-tile kernel
    Mr, Nr, Kr = 700, 512, 384
    ar = rng.integers(-128, 128, (Mr, Kr)).astype(np.int8)
    br = rng.integers(-128, 128, (Nr, Kr)).astype(np.int8)
    biasr = rng.integers(-30000, 30001, Nr).astype(np.int32)
    for relu in (False, True):
        for bs in (None, biasr):
            e, name = _run_i8(ar, br, Mr, Nr, Kr, 1e-3, bias=bs, relu=relu, flags=capi.FLAG_TILE_256)
            assert name == "wq_gemm_i8_mfma_256"
            ref = port.gemm_i8_bias_requant(ar, br, bs, Mr, Nr, Kr, Kr, Kr, Nr, 1e-3, int(relu)).reshape(Mr, Nr)
            np.testing.assert_array_equal(e, ref)
    big = 4096
    ab = rng.integers(-128, 128, (big, big)).astype(np.int8)
    bb = rng.integers(-128, 128, (big, big)).astype(np.int8)
    e256, name = _run_i8(ab, bb, big, big, big, 5e-5)
    assert name == "wq_gemm_i8_mfma_256"
    e128, name = _run_i8(ab, bb, big, big, big, 5e-5, flags=capi.FLAG_TILE_128)
    assert name == "wq_gemm_i8_mfma_128"
    np.testing.assert_array_equal(e256, e128)
    assert np.unique(e256).size > 50
    # a missing bias pointer is a malformed problem
    p = capi.fpAintB_problem(1, 1, None, 1, 128, 128, 128, a_dtype=capi.I8, b_dtype=capi.I8, c_dtype=capi.I8,
                             epilogue=capi.EPI_ADD_RELU_MUL_CLAMP, requant=0.03)
    assert not capi.is_supported(p)


def test_f8_int4_multiply_multiply_generic(golden_gemm):
    """Config C5 semantics (fp8 FNUZ bytes x int4, d0[n] * d1[m], optional FastGelu), golden case."""
    g = golden_gemm
    M, N, K = 48, 64, 128
    bp = pack_on_device(np.ascontiguousarray(g["f8_b"]), K, N)
    for key, kind in (("f8_e_mm", capi.EPI_MULTIPLY_MULTIPLY), ("f8_e_mm_gelu", capi.EPI_MULTIPLY_MULTIPLY_FASTGELU)):
        got = run_fpaintb(g["f8_a"], bp, None, M, N, K, a_dtype=capi.F8_FNUZ, epilogue=kind,
                          d0=g["f8_d0"], d1=g["f8_d1"], expect_kernel=GENERIC)
        exp = g[key].reshape(M, N)
        if kind == capi.EPI_MULTIPLY_MULTIPLY:
            np.testing.assert_array_equal(got, exp)
        else:
            check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


F8_MFMA = "wq_gemm_f8_mfma_256"


def _f8_bytes(rng, shape, exclude=(0x7F, 0xFF, 0x80)):
    """Uniform random fp8 bytes without the given encodings."""
    allowed = np.array([b for b in range(256) if b not in exclude], dtype=np.uint8)
    return allowed[rng.integers(0, allowed.size, shape)]


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (256, 512, 384), (512, 256, 1024), (300, 256, 512), (100, 256, 256)])
def test_f8_mfma_fnuz_vs_checker(port, M, N, K):
    """fp8 FNUZ bytes x packed int4 on the fp8 matrix core (caller excludes 0x7F/0xFF/0x80), MultiplyMultiply
    [+ FastGelu], against the CPU checker and the generic kernel."""
    rng = np.random.default_rng(M + N + K)
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    d0 = rng.uniform(0.0, 1.0, N).astype(np.float32)
    d1 = rng.uniform(0.0, 1.0, M).astype(np.float32)
    # exact set: activations in {-2..2} (bytes of exactly representable integers), power-of-two scales
    lut = {0: 0x00, 1: 0x40, 2: 0x48, -1: 0xC0, -2: 0xC8}
    a_int = rng.integers(-2, 3, (M, K))
    a = np.vectorize(lut.get)(a_int).astype(np.uint8)
    d0e = (2.0 ** rng.integers(-3, 1, N)).astype(np.float32)
    d1e = (2.0 ** rng.integers(-3, 1, M)).astype(np.float32)
    got = run_fpaintb(a, bp, None, M, N, K, a_dtype=capi.F8_FNUZ, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0e, d1=d1e,
                      flags=capi.FLAG_F8_NO_SPECIALS, expect_kernel=F8_MFMA)
    exp = port.gemm_f8_i8_mm(a, q, d0e, d1e, M, N, K, K, K, N, False).reshape(M, N)
    np.testing.assert_array_equal(got, exp)
    # realistic set: activations ~ U(-1, 1) rounded to fp8 FNUZ (the value range of the reference's own
    # GeneratorTensor_3 inputs, example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8.cpp:170-190):
    # check_err defaults for a half output
    a = torch.from_numpy(rng.uniform(-1, 1, (M, K)).astype(np.float32)).to(torch.float8_e4m3fnuz).view(torch.uint8).numpy()
    a[(a == 0x80)] = 0
    for gelu, kind in ((False, capi.EPI_MULTIPLY_MULTIPLY), (True, capi.EPI_MULTIPLY_MULTIPLY_FASTGELU)):
        got = run_fpaintb(a, bp, None, M, N, K, a_dtype=capi.F8_FNUZ, epilogue=kind, d0=d0, d1=d1,
                          flags=capi.FLAG_F8_NO_SPECIALS, expect_kernel=F8_MFMA)
        exp = port.gemm_f8_i8_mm(a, q, d0, d1, M, N, K, K, K, N, gelu).reshape(M, N)
        check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
        gen = run_fpaintb(a, bp, None, M, N, K, a_dtype=capi.F8_FNUZ, epilogue=kind, d0=d0, d1=d1,
                          flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
        if not gelu:
            np.testing.assert_array_equal(gen, exp)
    # adversarial set: uniformly random bytes (magnitudes log-uniform over 2^-10 .. 224).  The fp8 matrix core
    # aligns the products of one lane's 32 k to the largest of them and drops what lies more than ~14 bits
    # below it, so the stated bound gains a term relative to sum_k |a q|: 2^-12 * sum|a q| * d0 * d1.
    a = _f8_bytes(rng, (M, K))
    got = bits_to_f32(run_fpaintb(a, bp, None, M, N, K, a_dtype=capi.F8_FNUZ, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0,
                                  d1=d1, flags=capi.FLAG_F8_NO_SPECIALS, expect_kernel=F8_MFMA), capi.F16)
    exp = bits_to_f32(port.gemm_f8_i8_mm(a, q, d0, d1, M, N, K, K, K, N, False).reshape(M, N), capi.F16)
    a_abs = np.abs(torch.from_numpy(a).view(torch.float8_e4m3fnuz).to(torch.float64).numpy())
    terms = (a_abs @ np.abs(q.astype(np.float64)).T) * d0[None, :] * d1[:, None]
    assert (np.abs(got.astype(np.float64) - exp) <= 1e-3 + 1e-3 * np.abs(exp) + 2.0 ** -12 * terms).all()
    # without the caller's promise FNUZ data is converted to fp16 (exact for every byte) and runs on the fp16
    # matrix-core kernel: all 256 encodings, +-240 included, within check_err defaults; NaN poisons its row
    a = rng.integers(0, 256, (M, K)).astype(np.uint8)
    a[a == 0x80] = 0x7F
    a[5, 17] = 0x80
    exp = bits_to_f32(port.gemm_f8_i8_mm(a, q, d0, d1, M, N, K, K, K, N, False).reshape(M, N), capi.F16)
    ok = np.ones(M, bool)
    ok[5] = False
    for flags, tile in ((0, 128), (capi.FLAG_TILE_256, 256)):
        got = bits_to_f32(run_fpaintb(a, bp, None, M, N, K, a_dtype=capi.F8_FNUZ, epilogue=capi.EPI_MULTIPLY_MULTIPLY,
                                      d0=d0, d1=d1, flags=flags,
                                      expect_kernel=f"wq_f8_fnuz_to_f16+wq_gemm_mfma_{tile}"), capi.F16)
        assert np.isnan(got[5]).all() and np.isnan(exp[5]).all()
        assert not np.isnan(got[ok]).any()
        check_err(got[ok], exp[ok], 1e-3, 1e-3)


def test_f8_mfma_e4m3fn(port):
    """E4M3FN (OCP) bytes: every encoding is native to the matrix core.  Finite bytes decode to twice the FNUZ
    value of the same byte, so the FNUZ checker with d1 doubled is the exact expectation; NaN bytes (0x7F, 0xFF)
    poison their row; the generic kernel and torch's float8_e4m3fn decode agree."""
    rng = np.random.default_rng(65)
    M, N, K = 256, 256, 512
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    d0 = rng.uniform(0.0, 1.0, N).astype(np.float32)
    d1 = rng.uniform(0.0, 1.0, M).astype(np.float32)
    a = torch.from_numpy(rng.uniform(-1, 1, (M, K)).astype(np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    a[(a == 0x80)] = 0
    got = run_fpaintb(a, bp, None, M, N, K, a_dtype=capi.F8_E4M3FN, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0, d1=d1,
                      expect_kernel=F8_MFMA)
    exp = port.gemm_f8_i8_mm(a, q, d0, 2.0 * d1, M, N, K, K, K, N, False).reshape(M, N)
    check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
    gen = run_fpaintb(a, bp, None, M, N, K, a_dtype=capi.F8_E4M3FN, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0, d1=d1,
                      flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
    np.testing.assert_array_equal(gen, exp)
    # independent decode: torch.float8_e4m3fn on the CPU, float64 accumulation
    a_f = torch.from_numpy(a).view(torch.float8_e4m3fn).to(torch.float64)
    ref = (a_f @ torch.from_numpy(q).to(torch.float64).T) * torch.from_numpy(d0).double()[None, :] \
        * torch.from_numpy(d1).double()[:, None]
    check_err(bits_to_f32(got, capi.F16), ref.numpy().astype(np.float32), 2e-3, 2e-3)
    # -0 (0x80) is an ordinary zero, NaN bytes poison exactly their rows
    a2 = a.copy()
    a2[3, 5] = 0x80
    a2[7, 100] = 0x7F
    a2[200, 511] = 0xFF
    got2 = bits_to_f32(run_fpaintb(a2, bp, None, M, N, K, a_dtype=capi.F8_E4M3FN, epilogue=capi.EPI_MULTIPLY_MULTIPLY,
                                   d0=d0, d1=d1, expect_kernel=F8_MFMA), capi.F16)
    gen2 = bits_to_f32(run_fpaintb(a2, bp, None, M, N, K, a_dtype=capi.F8_E4M3FN, epilogue=capi.EPI_MULTIPLY_MULTIPLY,
                                   d0=d0, d1=d1, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC), capi.F16)
    nan_rows = np.isnan(got2).all(axis=1)
    assert nan_rows[7] and nan_rows[200] and nan_rows.sum() == 2
    assert (np.isnan(gen2).all(axis=1) == nan_rows).all()
    ok = ~nan_rows
    check_err(got2[ok], gen2[ok], 1e-3, 1e-3)


@pytest.mark.parametrize("k_batch", [2, 3, 5, 8])
def test_splitk_in_gpu(port, k_batch):
    """k_batches of the reference's split-K tests (test/gemm_split_k/test_gemm_splitk_util.hpp:36-87)."""
    rng = np.random.default_rng(k_batch)
    M, N, K = 128, 128, 1024
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = f16_bits(rng.integers(1, 3, N).astype(np.float32))
    got = run_fpaintb(a, bp, s, M, N, K, k_batch=k_batch, expect_kernel=MFMA)
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    np.testing.assert_array_equal(got, exp)
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    got = run_fpaintb(a, bp, s, M, N, K, k_batch=k_batch, expect_kernel=MFMA)
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


def test_k_slices_then_reduce_finalize(port):
    """The multi-GPU split-K data path on one GPU: G slice launches -> fp32 partials -> sum -> finalize."""
    rng = np.random.default_rng(21)
    M, N, K, G = 128, 256, 1024, 4
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = f16_bits(rng.integers(1, 3, N).astype(np.float32))
    a_t, s_t = dev(a), dev(s)
    parts = torch.zeros((G, M, N), dtype=torch.float32, device="cuda")
    kread = capi.splitk_kread(K, G)
    stream = torch.cuda.current_stream().cuda_stream
    for gidx in range(G):
        p = capi.fpAintB_problem(a_t.data_ptr(), bp.data_ptr(), s_t.data_ptr(), parts[gidx].data_ptr(), M, N, K,
                                 k_begin=gidx * kread, k_end=min(K, (gidx + 1) * kread), c_dtype=capi.F32)
        assert capi.kernel_name(p) == MFMA
        capi.launch(p, None, stream)
    torch.cuda.synchronize()
    # each partial equals the checker's partial exactly (integer inputs)
    for gidx in range(G):
        exp = port.fpAintB_partial(a, q, s, M, N, K, K, gidx * kread, min(K, (gidx + 1) * kread))
        np.testing.assert_array_equal(parts[gidx].cpu().numpy(), exp)
    total = parts.sum(dim=0).contiguous()      # stands in for the RCCL reduce
    c_t = torch.zeros((M, N), dtype=torch.int16, device="cuda")
    p = capi.fpAintB_problem(a_t.data_ptr(), bp.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K)
    capi.reduce_finalize(p, total.data_ptr(), 1, stream)
    torch.cuda.synchronize()
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    np.testing.assert_array_equal(host_u16(c_t), exp)
    # reduce_finalize summing the slabs itself (k_batch = G)
    c_t.zero_()
    capi.reduce_finalize(p, parts.data_ptr(), G, stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host_u16(c_t), exp)


def test_full_size_c2_properties(port):
    """BASELINE config 2 at full size (M=N=K=4096): a 32-row block against the CPU checker, the
    whole output against the independent generic HIP kernel, and exactness on integer inputs."""
    rng = np.random.default_rng(4096)
    M = N = K = 4096
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    got = run_fpaintb(a, bp, s, M, N, K, expect_kernel=MFMA256)
    rows = slice(1000, 1032)
    exp = port.fpAintB_gemm_f16(np.ascontiguousarray(a[rows]), q, s, 32, N, K, K, K, N).reshape(32, N)
    check_err(bits_to_f32(got[rows], capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
    # independent HIP path: generic kernel (bit-exact with the checker at small sizes) on a 96-row block; the same
    # block through the dispatcher's own choice for a short, ragged M (128-tile kernel, planner-chosen split-K)
    a96 = np.ascontiguousarray(a[2048:2048 + 96])
    gen = run_fpaintb(a96, bp, s, 96, N, K, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
    check_err(bits_to_f32(got[2048:2048 + 96], capi.F16), bits_to_f32(gen, capi.F16), 1e-3, 1e-3)
    rag = run_fpaintb(a96, bp, s, 96, N, K, expect_kernel=MFMA)
    check_err(bits_to_f32(rag, capi.F16), bits_to_f32(gen, capi.F16), 1e-3, 1e-3)
    # integer-valued inputs at full size: the MFMA kernel must equal the generic kernel bit for bit
    ai = f16_bits(np.round(rng.uniform(-1, 1, (M, K))))
    si = f16_bits(rng.integers(1, 3, N).astype(np.float32))
    got = run_fpaintb(ai, bp, si, M, N, K, expect_kernel=MFMA256)
    gen = run_fpaintb(np.ascontiguousarray(ai[512:512 + 96]), bp, si, 96, N, K, flags=capi.FLAG_GENERIC,
                      expect_kernel=GENERIC)
    # ... and the 128-tile kernel on the same full-size problem
    got128 = run_fpaintb(ai, bp, si, M, N, K, flags=capi.FLAG_TILE_128, expect_kernel=MFMA)
    np.testing.assert_array_equal(got128, got)
    np.testing.assert_array_equal(got[512:512 + 96], gen)
    # linearity in the scale: doubling every scale doubles the output exactly (power of two)
    got2 = run_fpaintb(ai, bp, f16_bits(2 * si.view(np.float16).astype(np.float32)), M, N, K)
    np.testing.assert_array_equal(bits_to_f32(got2, capi.F16), 2 * bits_to_f32(got, capi.F16))


def _launch(p, ws=None):
    capi.launch(p, ws.data_ptr() if ws is not None else None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()


def test_full_size_c3_properties(port):
    """BASELINE config 3 at full size (bf16 x int8, M = 4096, N = 11008, K = 4096): a row block against the CPU
    checker, exact agreement of the 256- and 128-tile kernels on integer inputs, exact linearity in the scale."""
    M, N, K = 4096, 11008, 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    b = torch.randint(-128, 128, (N, K), generator=g, device="cuda", dtype=torch.int8)
    c = torch.empty((M, N), dtype=torch.int16, device="cuda")

    def run(a, s, flags=0, expect=MFMA256):
        p = capi.fpAintB_problem(a.data_ptr(), b.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, a_dtype=capi.BF16,
                                 b_dtype=capi.I8, flags=flags)
        assert capi.kernel_name(p) == expect
        _launch(p)
        return c.clone()

    a = (torch.rand((M, K), generator=g, device="cuda") * 2 - 1).bfloat16()
    s = (torch.rand((N,), generator=g, device="cuda") * (2.0 ** -6 - 2.0 ** -10) + 2.0 ** -10).bfloat16()
    got = run(a, s)
    rows = slice(3000, 3016)
    a_h = a[rows].view(torch.int16).cpu().numpy().view(np.uint16)
    s_h = s.view(torch.int16).cpu().numpy().view(np.uint16)
    exp = port.gemm_bf16_i8(np.ascontiguousarray(a_h), b.cpu().numpy(), s_h, 16, N, K, K, 1, K, N, 0).reshape(16, N)
    got_rows = got[rows].cpu().numpy().view(np.uint16)
    check_err(bits_to_f32(got_rows, capi.BF16), bits_to_f32(exp, capi.BF16), 1e-2, 1e-2)
    # element-wise: the output is a truncation of an fp32 sum whose order differs from the checker's, so the two can
    # sit on either side of a bf16 boundary but no further -- at most one bf16 step apart wherever the value is large
    # against the fp32 summation error (|c| >= 1/8 here; below that the same absolute error spans several tiny steps)
    def ordered(bits):   # sign-magnitude bf16 patterns -> integers in numeric order
        b = bits.astype(np.int32)
        return np.where(b & 0x8000, -(b & 0x7FFF), b)
    big = np.abs(bits_to_f32(exp, capi.BF16)) >= 0.125
    assert big.mean() > 0.9
    assert np.abs(ordered(got_rows) - ordered(exp))[big].max() <= 1
    assert np.abs(bits_to_f32(got_rows, capi.BF16) - bits_to_f32(exp, capi.BF16))[~big].max() <= 2.0 ** -9
    # integer inputs: both tile kernels and the checker agree bit for bit
    ai = torch.randint(-1, 2, (M, K), generator=g, device="cuda").bfloat16()
    si = torch.randint(1, 3, (N,), generator=g, device="cuda").bfloat16()
    g256 = run(ai, si)
    g128 = run(ai, si, flags=capi.FLAG_TILE_128, expect=MFMA)
    assert torch.equal(g256, g128)
    exp = port.gemm_bf16_i8(np.ascontiguousarray(ai[rows].view(torch.int16).cpu().numpy().view(np.uint16)), b.cpu().numpy(),
                            si.view(torch.int16).cpu().numpy().view(np.uint16), 16, N, K, K, 1, K, N, 0).reshape(16, N)
    np.testing.assert_array_equal(g256[rows].cpu().numpy().view(np.uint16), exp)
    # doubling every scale doubles every output exactly
    g2 = run(ai, (si.float() * 2).bfloat16())
    assert torch.equal(g2.view(torch.bfloat16).float(), 2 * g256.view(torch.bfloat16).float())


def test_full_size_c4_slices_properties(port):
    """BASELINE config 4 on one GPU (fp16 x int4, M = N = 4096, K = 65536 in 8 slices of 8192): on integer inputs the sum
    of the eight fp32 slice partials + finalize equals the single-pass kernel over the whole K bit for bit; a row block
    of a slice partial equals the checker's partial; realistic inputs stay within the K-scaled tolerance."""
    M = N = 4096
    K, G = 65536, 8
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    q = torch.randint(-8, 8, (N, K), generator=g, device="cuda", dtype=torch.int8)
    bp = torch.empty((N, K // 2), dtype=torch.uint8, device="cuda")
    capi.pack_int4(q.data_ptr(), K, K, N, bp.data_ptr(), K // 2, torch.cuda.current_stream().cuda_stream)
    ai = torch.randint(-1, 2, (M, K), generator=g, device="cuda").half()
    si = torch.randint(1, 3, (N,), generator=g, device="cuda").half()
    kread = capi.splitk_kread(K, G)
    assert kread == 8192
    total = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    part = torch.empty((M, N), dtype=torch.float32, device="cuda")
    first = None
    for gi in range(G):
        p = capi.fpAintB_problem(ai.data_ptr(), bp.data_ptr(), si.data_ptr(), part.data_ptr(), M, N, K,
                                 k_begin=gi * kread, k_end=(gi + 1) * kread, c_dtype=capi.F32)
        assert capi.kernel_name(p) == MFMA256
        _launch(p)
        total += part
        if gi == 0:
            first = part[:8].cpu().numpy()
    exp = port.fpAintB_partial(ai[:8].cpu().numpy().view(np.uint16), q.cpu().numpy(), si.cpu().numpy().view(np.uint16), 8, N, K,
                               K, 0, kread)
    np.testing.assert_array_equal(first, exp.reshape(8, N))
    c_sum = torch.empty((M, N), dtype=torch.int16, device="cuda")
    c_one = torch.empty((M, N), dtype=torch.int16, device="cuda")
    p = capi.fpAintB_problem(ai.data_ptr(), bp.data_ptr(), si.data_ptr(), c_sum.data_ptr(), M, N, K)
    capi.reduce_finalize(p, total.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    p1 = capi.fpAintB_problem(ai.data_ptr(), bp.data_ptr(), si.data_ptr(), c_one.data_ptr(), M, N, K)
    assert capi.kernel_name(p1) == MFMA256
    _launch(p1)
    assert torch.equal(c_sum, c_one)
    # realistic inputs: slices + sum against the single pass; the fp32 summation order differs, so the tolerance of
    # check_err's half defaults is widened by sqrt(K / 4096)
    a = (torch.rand((M, K), generator=g, device="cuda") * 2 - 1).half()
    s = (torch.rand((N,), generator=g, device="cuda") * (2.0 ** -4 - 2.0 ** -8) + 2.0 ** -8).half()
    total.zero_()
    for gi in range(G):
        p = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), s.data_ptr(), part.data_ptr(), M, N, K,
                                 k_begin=gi * kread, k_end=(gi + 1) * kread, c_dtype=capi.F32)
        _launch(p)
        total += part
    p = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), s.data_ptr(), c_sum.data_ptr(), M, N, K)
    capi.reduce_finalize(p, total.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    p1 = capi.fpAintB_problem(a.data_ptr(), bp.data_ptr(), s.data_ptr(), c_one.data_ptr(), M, N, K)
    _launch(p1)
    tol = 1e-3 * (K / 4096) ** 0.5
    check_err(c_sum.view(torch.float16).float().cpu().numpy(), c_one.view(torch.float16).float().cpu().numpy(), tol, tol)
    exp = port.fpAintB_gemm_f16(a[:4].cpu().numpy().view(np.uint16), q.cpu().numpy(), s.cpu().numpy().view(np.uint16), 4, N, K,
                                K, K, N).reshape(4, N)
    check_err(c_one[:4].view(torch.float16).float().cpu().numpy(), bits_to_f32(exp, capi.F16), tol, tol)


def test_full_size_c5_properties(port):
    """BASELINE config 5 at full size (fp8 x int4 + MultiplyMultiply [+ FastGelu], 4096^3 and example/65's default
    3840 x 4096 x 4096): a row block against the CPU checker, exactness on the exact input set, and exact doubling of
    the output when the per-row factors double."""
    for M in (4096, 3840):
        N = K = 4096
        g = torch.Generator(device="cuda")
        g.manual_seed(5 + M)
        q = torch.randint(-8, 8, (N, K), generator=g, device="cuda", dtype=torch.int8)
        bp = torch.empty((N, K // 2), dtype=torch.uint8, device="cuda")
        capi.pack_int4(q.data_ptr(), K, K, N, bp.data_ptr(), K // 2, torch.cuda.current_stream().cuda_stream)
        a = (torch.rand((M, K), generator=g, device="cuda") * 2 - 1).to(torch.float8_e4m3fnuz).view(torch.uint8)
        a[a == 0x80] = 0
        d0 = torch.rand((N,), generator=g, device="cuda") * 2.0 ** -3
        d1 = torch.rand((M,), generator=g, device="cuda")
        c = torch.empty((M, N), dtype=torch.int16, device="cuda")

        def run(a_, d0_, d1_, kind):
            p = capi.fpAintB_problem(a_.data_ptr(), bp.data_ptr(), None, c.data_ptr(), M, N, K, a_dtype=capi.F8_FNUZ,
                                     c_dtype=capi.F16, epilogue=kind, d0=d0_.data_ptr(), d1=d1_.data_ptr(),
                                     flags=capi.FLAG_F8_NO_SPECIALS)
            assert capi.kernel_name(p) == F8_MFMA
            _launch(p)
            return c.clone()

        rows = slice(M - 16, M)
        q_h = q.cpu().numpy()
        for gelu, kind in ((False, capi.EPI_MULTIPLY_MULTIPLY), (True, capi.EPI_MULTIPLY_MULTIPLY_FASTGELU)):
            got = run(a, d0, d1, kind)
            exp = port.gemm_f8_i8_mm(np.ascontiguousarray(a[rows].cpu().numpy()), q_h, d0.cpu().numpy(),
                                     np.ascontiguousarray(d1[rows].cpu().numpy()), 16, N, K, K, K, N, gelu).reshape(16, N)
            check_err(got[rows].view(torch.float16).float().cpu().numpy(), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
        # exact set: activations in {-2 .. 2}, power-of-two factors: bit-exact against the checker
        lut = torch.tensor([0xC8, 0xC0, 0x00, 0x40, 0x48], dtype=torch.uint8, device="cuda")
        ae = lut[torch.randint(0, 5, (M, K), generator=g, device="cuda")]
        d0e = torch.pow(2.0, torch.randint(-3, 1, (N,), generator=g, device="cuda").float())
        d1e = torch.pow(2.0, torch.randint(-3, 1, (M,), generator=g, device="cuda").float())
        got = run(ae, d0e, d1e, capi.EPI_MULTIPLY_MULTIPLY)
        exp = port.gemm_f8_i8_mm(np.ascontiguousarray(ae[rows].cpu().numpy()), q_h, d0e.cpu().numpy(),
                                 np.ascontiguousarray(d1e[rows].cpu().numpy()), 16, N, K, K, K, N, False).reshape(16, N)
        np.testing.assert_array_equal(got[rows].cpu().numpy().view(np.uint16), exp)
        got2 = run(ae, d0e, 2 * d1e, capi.EPI_MULTIPLY_MULTIPLY)
        assert torch.equal(got2.view(torch.float16).float(), 2 * got.view(torch.float16).float())


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (256, 512, 192), (512, 256, 1024), (768, 1024, 4096)])
def test_tile256_kernel_fp16_int4(port, M, N, K):
    """The 256 x 256 kernel (3-buffer LDS ring, counted waits) forced at small and ragged tile counts:
    1, 2, 3 and many K tiles exercise the prologue / steady state / drain of the ring."""
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = f16_bits(rng.integers(1, 4, N).astype(np.float32))
    got = run_fpaintb(a, bp, s, M, N, K, flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
    if M * N * K <= 256 * 512 * 1024:
        exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    else:
        exp = run_fpaintb(a, bp, s, M, N, K, flags=capi.FLAG_TILE_128, expect_kernel=MFMA)
    np.testing.assert_array_equal(got, exp)
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    got = run_fpaintb(a, bp, s, M, N, K, flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
    if M * N * K <= 256 * 512 * 1024:
        exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    else:
        exp = run_fpaintb(a, bp, s, M, N, K, flags=capi.FLAG_TILE_128, expect_kernel=MFMA)
    check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


@pytest.mark.parametrize("a_dtype,b_dtype", [(capi.F16, capi.I8), (capi.F16, capi.U8_OFFSET), (capi.BF16, capi.I8),
                                             (capi.BF16, capi.I4_PACKED)])
def test_tile256_kernel_other_formats(port, a_dtype, b_dtype):
    rng = np.random.default_rng(a_dtype * 10 + b_dtype)
    M, N, K = 256, 512, 320
    lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
    q = rng.integers(lo, hi, (N, K)).astype(np.int8)
    if b_dtype == capi.I4_PACKED:
        b = pack_on_device(q, K, N)
    elif b_dtype == capi.U8_OFFSET:
        b = (q.astype(np.int16) + 128).astype(np.uint8)
    else:
        b = q
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits
    a = cvt(np.round(rng.uniform(-1, 1, (M, K))))
    s = cvt(rng.integers(1, 3, N).astype(np.float32))
    got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, flags=capi.FLAG_TILE_256,
                      expect_kernel=MFMA256)
    if a_dtype == capi.F16:
        exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    else:
        exp = port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("variant", [capi.FLAG_LOCKSTEP, capi.FLAG_WAVES_2X4, capi.FLAG_DEQUANT_IN_L,
                                     capi.FLAG_WAVES_2X4 | capi.FLAG_DEQUANT_IN_L, capi.FLAG_DIRECT_STORE,
                                     capi.FLAG_GROUP8])
def test_tile256_schedule_variants_agree(port, variant):
    """Every wave-grid / schedule variant of the 256-tile kernel (default: 1 x 8 waves, ping-pong, DMA in
    the MFMA section) accumulates each output in the same k order, so all give the same bits."""
    rng = np.random.default_rng(31)
    M, N, K = 512, 512, 704
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    base = run_fpaintb(a, bp, s, M, N, K, flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
    other = run_fpaintb(a, bp, s, M, N, K, flags=capi.FLAG_TILE_256 | variant, expect_kernel=MFMA256)
    np.testing.assert_array_equal(base, other)
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    check_err(bits_to_f32(base, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


@pytest.mark.parametrize("a_dtype", [capi.F16, capi.BF16])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 512, 192), (512, 256, 1024)])
def test_tile256_mfma_32x32_variant(port, a_dtype, M, N, K):
    """The 256-tile kernel on v_mfma_f32_32x32x16 (different k-to-lane mapping, accumulator layout, swizzles):
    bit-exact on integer data, check_err defaults on realistic data, all epilogue kinds and split-K."""
    rng = np.random.default_rng(M + 3 * N + 7 * K + a_dtype)
    flags = capi.FLAG_TILE_256 | capi.FLAG_MFMA_32X32
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits

    def ref(a, s):
        if a_dtype == capi.F16:
            return port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
        return port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)

    a = cvt(np.round(rng.uniform(-2, 2, (M, K))))
    s = cvt(rng.integers(1, 4, N).astype(np.float32))
    got = run_fpaintb(a, bp, s, M, N, K, a_dtype=a_dtype, flags=flags, expect_kernel=MFMA256)
    np.testing.assert_array_equal(got, ref(a, s))
    if K >= 192:
        got = run_fpaintb(a, bp, s, M, N, K, a_dtype=a_dtype, flags=flags, k_batch=3, expect_kernel=MFMA256)
        np.testing.assert_array_equal(got, ref(a, s))
    a = cvt(rng.uniform(-1, 1, (M, K)))
    s = cvt(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    got = run_fpaintb(a, bp, s, M, N, K, a_dtype=a_dtype, flags=flags, expect_kernel=MFMA256)
    tol = 1e-3 if a_dtype == capi.F16 else 1e-2
    check_err(bits_to_f32(got, a_dtype), bits_to_f32(ref(a, s), a_dtype), tol, 1e-3)
    # FastGelu epilogue against the default kernel (same epilogue code, different accumulator layout)
    g1 = run_fpaintb(a, bp, s, M, N, K, a_dtype=a_dtype, flags=flags, epilogue=capi.EPI_FASTGELU, expect_kernel=MFMA256)
    g0 = run_fpaintb(a, bp, s, M, N, K, a_dtype=a_dtype, flags=capi.FLAG_TILE_256, epilogue=capi.EPI_FASTGELU,
                     expect_kernel=MFMA256)
    check_err(bits_to_f32(g1, a_dtype), bits_to_f32(g0, a_dtype), tol, 1e-3)


def test_tile256_splitk_and_slices(port):
    rng = np.random.default_rng(77)
    M, N, K = 256, 256, 1024
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = f16_bits(rng.integers(1, 3, N).astype(np.float32))
    exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
    for kb in (2, 3, 5):
        got = run_fpaintb(a, bp, s, M, N, K, k_batch=kb, flags=capi.FLAG_TILE_256, expect_kernel=MFMA256)
        np.testing.assert_array_equal(got, exp)


SKINNY = "wq_gemm_skinny"


@pytest.mark.parametrize("M", [1, 5, 16, 17, 48, 64])
@pytest.mark.parametrize("a_dtype,b_dtype", [(capi.F16, capi.I4_PACKED), (capi.F16, capi.U8_OFFSET), (capi.BF16, capi.I8),
                                             (capi.BF16, capi.I4_PACKED)])
def test_small_m_kernel(port, M, a_dtype, b_dtype):
    """Decode-like shapes (M <= 64): the streaming small-M kernel with the planner's split-K (workspace given),
    as a single slice (no workspace), with a caller's k_batch and with an epilogue; rows beyond M stay untouched."""
    rng = np.random.default_rng(1000 * M + 10 * a_dtype + b_dtype)
    N, K = 192, 1024
    lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
    q = rng.integers(lo, hi, (N, K)).astype(np.int8)
    if b_dtype == capi.I4_PACKED:
        b = pack_on_device(q, K, N)
    elif b_dtype == capi.U8_OFFSET:
        b = (q.astype(np.int16) + 128).astype(np.uint8)
    else:
        b = q
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits

    def ref(a, s, ep=0, d0=None, d1=None):
        if a_dtype == capi.F16:
            return port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
        return port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, ep, d0, d1).reshape(M, N)

    a = cvt(np.round(rng.uniform(-2, 2, (M, K))))
    s = cvt(rng.integers(1, 3, N).astype(np.float32))
    exp = ref(a, s)
    for kw in (dict(), dict(use_workspace=False), dict(k_batch=3)):
        got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, expect_kernel=SKINNY, **kw)
        np.testing.assert_array_equal(got, exp)
    a = cvt(rng.uniform(-1, 1, (M, K)))
    s = cvt(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    tol = 1e-3 if a_dtype == capi.F16 else 1e-2
    got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, expect_kernel=SKINNY)
    check_err(bits_to_f32(got, a_dtype), bits_to_f32(ref(a, s), a_dtype), tol, 1e-3)
    if a_dtype == capi.BF16:
        d0 = bf16_bits(rng.uniform(-0.5, 0.5, N))
        got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, epilogue=capi.EPI_ADD_FASTGELU, d0=d0,
                          expect_kernel=SKINNY)
        check_err(bits_to_f32(got, a_dtype), bits_to_f32(ref(a, s, 4, d0, d0), a_dtype), tol, 1e-3)
    # padded output rows (ldc > N) keep their poison
    got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, ldc=N + 8, expect_kernel=SKINNY)
    assert (got[:, N:] == 0x7BFF).all()
    check_err(bits_to_f32(got[:, :N], a_dtype), bits_to_f32(ref(a, s), a_dtype), tol, 1e-3)


@pytest.mark.parametrize("M", [65, 100, 128, 200, 333])
@pytest.mark.parametrize("a_dtype,b_dtype", [(capi.F16, capi.I4_PACKED), (capi.BF16, capi.I8), (capi.F16, capi.U8_OFFSET)])
def test_ragged_m_on_the_128_tile_kernel(port, M, a_dtype, b_dtype):
    """Any M > 64 runs on the 128-tile matrix-core kernel: rows past M in the last row tile are neither read from
    beyond A nor stored (poisoned rows after C stay intact).  Short grids get a planner-chosen split-K when a workspace
    is given and run as one slice without it."""
    rng = np.random.default_rng(31 * M + b_dtype)
    N, K = 256, 1024
    lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
    q = rng.integers(lo, hi, (N, K)).astype(np.int8)
    if b_dtype == capi.I4_PACKED:
        b = pack_on_device(q, K, N)
    elif b_dtype == capi.U8_OFFSET:
        b = (q.astype(np.int16) + 128).astype(np.uint8)
    else:
        b = q
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits

    def ref(a, s):
        if a_dtype == capi.F16:
            return port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
        return port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)

    a = cvt(np.round(rng.uniform(-2, 2, (M, K))))
    s = cvt(rng.integers(1, 3, N).astype(np.float32))
    exp = ref(a, s)
    for kw in (dict(), dict(use_workspace=False), dict(k_batch=3), dict(flags=capi.FLAG_TILE_128)):
        got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, expect_kernel=MFMA, **kw)
        np.testing.assert_array_equal(got, exp)
    a = cvt(rng.uniform(-1, 1, (M, K)))
    s = cvt(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    tol = 1e-3 if a_dtype == capi.F16 else 1e-2
    got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, epilogue=capi.EPI_FASTGELU, ldc=N + 8,
                      expect_kernel=MFMA)
    assert (got[:, N:] == 0x7BFF).all()
    gen = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, epilogue=capi.EPI_FASTGELU, flags=capi.FLAG_GENERIC,
                      expect_kernel=GENERIC)
    check_err(bits_to_f32(got[:, :N], a_dtype), bits_to_f32(gen, a_dtype), tol, 1e-3)


@pytest.mark.parametrize("M", [257, 300, 700])
@pytest.mark.parametrize("a_dtype,b_dtype", [(capi.F16, capi.I4_PACKED), (capi.BF16, capi.I8)])
def test_ragged_m_on_the_256_tile_kernel(port, M, a_dtype, b_dtype):
    """The 256-tile kernel with a partial last row tile (LDS-staged and direct-store epilogues, split-K): bit-exact on
    integer inputs, rows after C untouched."""
    rng = np.random.default_rng(17 * M + b_dtype)
    N, K = 512, 512
    lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
    q = rng.integers(lo, hi, (N, K)).astype(np.int8)
    b = pack_on_device(q, K, N) if b_dtype == capi.I4_PACKED else q
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits

    def ref(a, s):
        if a_dtype == capi.F16:
            return port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
        return port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)

    a = cvt(np.round(rng.uniform(-2, 2, (M, K))))
    s = cvt(rng.integers(1, 3, N).astype(np.float32))
    exp = ref(a, s)
    for kw in (dict(flags=capi.FLAG_TILE_256), dict(flags=capi.FLAG_TILE_256 | capi.FLAG_DIRECT_STORE),
               dict(flags=capi.FLAG_TILE_256, k_batch=2), dict(flags=capi.FLAG_TILE_256, ldc=N + 8)):
        got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, expect_kernel=MFMA256, **kw)
        if "ldc" in kw:
            assert (got[:, N:] == 0x7BFF).all()
            got = got[:, :N]
        np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("M,K", [(9, 8192), (33, 4480), (40, 8192), (64, 11008), (20, 192)])
@pytest.mark.parametrize("a_dtype,b_dtype", [(capi.F16, capi.I4_PACKED), (capi.BF16, capi.I8)])
def test_small_m_kernel_long_k(port, M, K, a_dtype, b_dtype):
    """Long and odd K ranges: the activations pass through LDS chunk by chunk, so K is not bounded by LDS; K that is
    not a whole number of 4-super-step groups runs with one load in flight; with and without a workspace."""
    if b_dtype == capi.I4_PACKED and K % 128:
        K = K // 128 * 128 + 128
    rng = np.random.default_rng(M * K + b_dtype)
    N = 128
    lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
    q = rng.integers(lo, hi, (N, K)).astype(np.int8)
    b = pack_on_device(q, K, N) if b_dtype == capi.I4_PACKED else q
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits

    def ref(a, s):
        if a_dtype == capi.F16:
            return port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
        return port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)

    a = cvt(np.round(rng.uniform(-2, 2, (M, K))))
    s = cvt(rng.integers(1, 3, N).astype(np.float32))
    exp = ref(a, s)
    for kw in (dict(), dict(use_workspace=False), dict(k_batch=2)):
        got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, expect_kernel=SKINNY, **kw)
        np.testing.assert_array_equal(got, exp)
    a = cvt(rng.uniform(-1, 1, (M, K)))
    s = cvt(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    tol = 1e-3 if a_dtype == capi.F16 else 1e-2
    got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, expect_kernel=SKINNY)
    check_err(bits_to_f32(got, a_dtype), bits_to_f32(ref(a, s), a_dtype), tol, 1e-3 * max(1.0, (K / 4096) ** 0.5))


@pytest.mark.parametrize("M", [1, 3, 9, 16])
@pytest.mark.parametrize("a_dtype,b_dtype,K", [(capi.F16, capi.I4_PACKED, 1024), (capi.F16, capi.I4_PACKED, 1536),
                                               (capi.F16, capi.I4_PACKED, 4096), (capi.BF16, capi.I8, 2048),
                                               (capi.F16, capi.I4_PACKED, 1152), (capi.F16, capi.I4_PACKED, 2816),
                                               (capi.BF16, capi.I8, 1344),
                                               (capi.BF16, capi.I8, 1024), (capi.BF16, capi.I4_PACKED, 512),
                                               (capi.F16, capi.U8_OFFSET, 768)])
def test_small_m_kernel_waves_split_k(port, M, a_dtype, b_dtype, K):
    """M <= 16 with a wide N: one kernel, the four waves of a workgroup split K (in units of 4, 2 or 1 super-steps,
    unevenly when the units do not divide by four) and sum through LDS."""
    rng = np.random.default_rng(7 * M + K + b_dtype)
    N = 2048
    lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
    q = rng.integers(lo, hi, (N, K)).astype(np.int8)
    if b_dtype == capi.I4_PACKED:
        b = pack_on_device(q, K, N)
    elif b_dtype == capi.U8_OFFSET:
        b = (q.astype(np.int16) + 128).astype(np.uint8)
    else:
        b = q
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits

    def ref(a, s):
        if a_dtype == capi.F16:
            return port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
        return port.gemm_bf16_i8(a, q, s, M, N, K, K, 1, K, N, 0).reshape(M, N)

    # the whole activation image in LDS up to 40 KiB, chunk by chunk through per-wave regions beyond
    name = "wq_gemm_skinny_ws" if M * (2 * K + 16) <= 40 * 1024 else "wq_gemm_skinny_wsc"
    a = cvt(np.round(rng.uniform(-2, 2, (M, K))))
    s = cvt(rng.integers(1, 3, N).astype(np.float32))
    got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, expect_kernel=name)
    np.testing.assert_array_equal(got, ref(a, s))
    a = cvt(rng.uniform(-1, 1, (M, K)))
    s = cvt(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    tol = 1e-3 if a_dtype == capi.F16 else 1e-2
    got = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, epilogue=capi.EPI_FASTGELU, ldc=N + 8,
                      expect_kernel=name)
    assert (got[:, N:] == 0x7BFF).all()
    gen = run_fpaintb(a, b, s, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype, epilogue=capi.EPI_FASTGELU, flags=capi.FLAG_GENERIC,
                      expect_kernel=GENERIC)
    check_err(bits_to_f32(got[:, :N], a_dtype), bits_to_f32(gen, a_dtype), tol, 1e-3)


@pytest.mark.parametrize("bits,dt", [(4, torch.float16), (8, torch.bfloat16)])
def test_torch_linear_over_packed_weights(tmp_path, bits, dt):
    """Quantise -> pack -> save -> load -> WqLinear (torch custom op over the C ABI) against a float64 matmul with the
    dequantised weights; ragged batch (rows padded to whole tiles inside the op) and a bias + FastGelu epilogue."""
    from composable_kernel_amd import weights
    from composable_kernel_amd.torch_ops import WqLinear
    rng = np.random.default_rng(bits)
    N, K = 384, 512
    sdt = "f16" if dt == torch.float16 else "bf16"
    w = rng.normal(0, 0.05, (N, K)).astype(np.float32)
    q, s = weights.quantize_per_channel(w, bits, sdt)
    path = str(tmp_path / "layer.npz")
    weights.save_packed(path, weights.pack(q, s, bits, sdt))
    pw = weights.load_packed(path)
    wd = torch.from_numpy(weights.dequantize(pw)).double()
    x = torch.from_numpy(rng.uniform(-1, 1, (3, 50, K)).astype(np.float32)).to(dt)
    tol = 2e-3 if dt == torch.float16 else 1.6e-2
    y = WqLinear(pw)(x.cuda())
    assert y.shape == (3, 50, N) and y.dtype == dt
    ref = x.double() @ wd.T
    assert torch.allclose(y.cpu().double(), ref, rtol=tol, atol=tol)
    bias = torch.from_numpy(rng.uniform(-0.5, 0.5, N).astype(np.float32))
    y = WqLinear(pw, bias=bias, epilogue="fastgelu")(x.cuda())
    z = ref + bias.to(dt).double()
    gelu = z / (1 + torch.exp(z * (-2 * 0.035677 * z * z - 2 * 0.797885)))
    assert torch.allclose(y.cpu().double(), gelu, rtol=tol, atol=tol)
    # a decode-sized batch takes the small-M kernel
    xs = x[0, :7].contiguous().cuda()
    ys = WqLinear(pw)(xs)
    assert torch.allclose(ys.cpu().double(), xs.cpu().double() @ wd.T, rtol=tol, atol=tol)
    # registered as a dispatcher operator (usable from torch.compile graphs through its fake implementation)
    assert torch.ops.wqgemm.fpaintb_linear is not None
    # operands that would make the kernels read past an allocation or reinterpret bytes are refused before any launch
    from composable_kernel_amd.torch_ops import fpaintb_linear
    lin = WqLinear(pw)
    xc = x[0].contiguous().cuda()
    for kw in (dict(bits=5), dict(weight=lin.weight[:, :-1].contiguous()), dict(weight=lin.weight[:-1]),
               dict(weight=lin.weight.cpu()), dict(bias=bias.cuda().float(), epilogue="add"),
               dict(bias=bias.to(dt).cuda()[:-1].contiguous(), epilogue="add"), dict(weight=lin.weight.t().contiguous().t())):
        args = dict(x=xc, weight=lin.weight, scale=lin.scale, bits=bits, epilogue="none", bias=None)
        args.update(kw)
        with pytest.raises((ValueError, TypeError, RuntimeError)):
            fpaintb_linear(args["x"], args["weight"], args["scale"], args["bits"], args["epilogue"], args["bias"])


def test_torch_linear_group_wise_scales(tmp_path):
    """Group-wise quantised weights through the container and the torch operator: closer to the float weights than
    per-channel scales on channels whose magnitude varies along k, and equal to a float64 matmul with the dequantised
    weights within the fp16 bar."""
    from composable_kernel_amd import weights
    from composable_kernel_amd.torch_ops import WqLinear
    rng = np.random.default_rng(128)
    N, K, group_k = 512, 1024, 128
    w = (rng.normal(0, 0.05, (N, K)) * np.linspace(0.2, 3.0, K)[None, :]).astype(np.float32)
    q, s = weights.quantize_per_group(w, group_k, 4, "f16")
    path = str(tmp_path / "grouped.npz")
    weights.save_packed(path, weights.pack(q, s, 4, "f16", group_k))
    pw = weights.load_packed(path)
    lin = WqLinear(pw)
    assert tuple(lin.scale.shape) == (K // group_k, N)
    wd = torch.from_numpy(weights.dequantize(pw)).double()
    for rows in (300, 5):
        x = torch.from_numpy(rng.uniform(-1, 1, (rows, K)).astype(np.float32)).half()
        y = lin(x.cuda())
        assert torch.allclose(y.cpu().double(), x.double() @ wd.T, rtol=2e-3, atol=2e-3)
    qc, sc = weights.quantize_per_channel(w, 4, "f16")
    err_g = (wd - torch.from_numpy(w).double()).abs().mean()
    err_c = (torch.from_numpy(weights.dequantize(weights.pack(qc, sc, 4, "f16"))).double() - torch.from_numpy(w).double()).abs().mean()
    assert err_g < 0.8 * err_c
    with pytest.raises((ValueError, TypeError, RuntimeError)):
        lin.scale = lin.scale[:3].contiguous()       # 1024 is not a whole number of three groups
        lin(x.cuda())


@pytest.mark.parametrize("layout", ["packed", "preshuffled"])
def test_torch_linear_group_wise_scales_bf16(tmp_path, layout):
    """The same through bf16 activations and 8-bit weights (the Multiply-on-B form with a scale row per group), for a
    prefill-sized, a chunk-sized (planner split-K) and a decode-sized batch, plain and pre-shuffled weight order:
    against a float64 matmul with the dequantised weights within the bf16 bar (one truncation of a sum of 1024 terms)."""
    from composable_kernel_amd import weights
    from composable_kernel_amd.torch_ops import WqLinear
    rng = np.random.default_rng(129)
    N, K, group_k = 512, 1024, 128
    w = (rng.normal(0, 0.05, (N, K)) * np.linspace(0.2, 3.0, K)[None, :]).astype(np.float32)
    q, s = weights.quantize_per_group(w, group_k, 8, "bf16")
    pw = weights.pack(q, s, 8, "bf16", group_k)
    if layout == "preshuffled":
        pw = weights.preshuffle(pw)
    path = str(tmp_path / "grouped_bf16.npz")
    weights.save_packed(path, pw)
    pw = weights.load_packed(path)
    assert pw.layout == layout
    lin = WqLinear(pw)
    wd = torch.from_numpy(weights.dequantize(pw)).double()
    for rows in (300, 40, 5):
        x = torch.from_numpy(rng.uniform(-1, 1, (rows, K)).astype(np.float32)).bfloat16()
        y = lin(x.cuda())
        assert y.dtype == torch.bfloat16
        assert torch.allclose(y.cpu().double(), x.double() @ wd.T, rtol=1e-2, atol=1e-2)


def test_launch_is_graph_capturable(port):
    """wqgemm_launch neither allocates nor synchronises, so a stream capture can record it: a small-M problem with the
    planner's split-K (two kernels and a workspace) and a 256-tile problem are captured into one HIP graph, replayed
    on fresh inputs, and must give the results of plain launches."""
    rng = np.random.default_rng(99)
    cases = []
    for (M, N, K) in ((8, 256, 2048), (256, 256, 512)):
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        bp = pack_on_device(q, K, N)
        a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
        s = f16_bits(rng.integers(1, 3, N).astype(np.float32))
        exp = port.fpAintB_gemm_f16(a, q, s, M, N, K, K, K, N).reshape(M, N)
        a_t, s_t = dev(np.zeros_like(a)), dev(s)
        c_t = torch.zeros((M, N), dtype=torch.int16, device="cuda")
        p = capi.fpAintB_problem(a_t.data_ptr(), bp.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K,
                                 flags=capi.FLAG_TILE_256 if M == 256 else 0)
        nbytes = capi.workspace_bytes(p)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
        cases.append((p, ws if nbytes else None, a, a_t, c_t, exp, bp, s_t))
    assert cases[0][1] is not None and capi.kernel_name(cases[0][0]) == SKINNY
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sp = torch.cuda.current_stream().cuda_stream
        for p, ws, *_ in cases:
            capi.launch(p, ws.data_ptr() if ws is not None else None, sp)
    for _, _, a, a_t, c_t, exp, *_ in cases:  # inputs arrive after the capture
        a_t.copy_(torch.from_numpy(a.view(np.int16)).cuda().view(a_t.dtype))
        c_t.zero_()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for _, _, a, a_t, c_t, exp, *_ in cases:
        np.testing.assert_array_equal(c_t.cpu().numpy().view(np.uint16), exp)


def test_random_shapes_every_route_against_the_general_kernel():
    """Seeded sweep over shapes, formats, epilogues and workspace use: whatever kernel the dispatcher picks must agree
    bit for bit with the general kernel (the checker's own summation order) on integer-valued inputs, and must not
    touch the guard rows behind C."""
    rng = np.random.default_rng(20251111)
    seen = {}
    for case in range(90):
        a_dtype = (capi.F16, capi.BF16)[int(rng.integers(0, 2))]
        b_dtype = (capi.I4_PACKED, capi.I8, capi.U8_OFFSET)[int(rng.integers(0, 3))]
        M = int(rng.choice([1, 2, 3, 7, 16, 17, 31, 33, 64, 65, 100, 128, 129, 255, 256, 257, 300, 511, 600]))
        N = int(rng.choice([16, 64, 100, 128, 192, 256, 384, 516, 768, 1024, 2048, 2048]))
        K = int(rng.choice([128, 256, 384, 512, 640, 1024, 1152, 2048]))
        lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
        q = rng.integers(lo, hi, (N, K)).astype(np.int8)
        if b_dtype == capi.I4_PACKED:
            b = pack_on_device(q, K, N)
        elif b_dtype == capi.U8_OFFSET:
            b = (q.astype(np.int16) + 128).astype(np.uint8)
        else:
            b = q
        cvt = f16_bits if a_dtype == capi.F16 else bf16_bits
        a = cvt(np.round(rng.uniform(-2, 2, (M, K))))
        s = cvt(rng.integers(1, 3, N).astype(np.float32))
        kw = dict(a_dtype=a_dtype, b_dtype=b_dtype)
        if rng.integers(0, 3) == 0:
            kw["use_workspace"] = False
        elif rng.integers(0, 3) == 0 and K >= 512:
            kw["k_batch"] = int(rng.integers(2, 5))
        if rng.integers(0, 2):
            kw["ldc"] = N + 8
        got = run_fpaintb(a, b, s, M, N, K, **kw)
        kw.pop("use_workspace", None)
        ref = run_fpaintb(a, b, s, M, N, K, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC, **kw)
        np.testing.assert_array_equal(got[:, :N], ref[:, :N], err_msg=f"case {case}: M={M} N={N} K={K} {kw}")
        if "ldc" in kw:
            assert (got[:, N:] == 0x7BFF).all()
        p = capi.fpAintB_problem(1 << 20, 1 << 20, 1 << 20, 1 << 20, M, N, K, a_dtype=a_dtype, b_dtype=b_dtype)
        seen[capi.kernel_name(p)] = seen.get(capi.kernel_name(p), 0) + 1
    # the sweep must have exercised every kernel family
    for name in ("wq_gemm_skinny_ws", "wq_gemm_skinny", "wq_gemm_mfma_128"):
        assert any(k.startswith(name) for k in seen), seen


def test_random_shapes_int8_requant_against_the_general_kernel():
    """The same sweep for the int8 requantisation family (tile kernels vs the general kernel, exact)."""
    rng = np.random.default_rng(1411)
    names = set()
    for case in range(40):
        M = int(rng.choice([1, 5, 64, 100, 128, 200, 256, 300, 513]))
        N = int(rng.choice([16, 100, 128, 256, 384, 512, 1024]))
        K = int(rng.choice([128, 256, 640, 1024]))
        a = rng.integers(-128, 128, (M, K)).astype(np.int8)
        b = rng.integers(-128, 128, (N, K)).astype(np.int8)
        bias = rng.integers(-50000, 50001, N).astype(np.int32) if rng.integers(0, 2) else None
        relu = bool(rng.integers(0, 2))
        rq = float(rng.choice([1e-4, 1e-3, 0.03]))
        flags = capi.FLAG_TILE_256 if (N % 256 == 0 and rng.integers(0, 2)) else 0
        e, name = _run_i8(a, b, M, N, K, rq, bias=bias, relu=relu, flags=flags)
        g, gname = _run_i8(a, b, M, N, K, rq, bias=bias, relu=relu, flags=capi.FLAG_GENERIC)
        assert gname == GENERIC and name != GENERIC
        names.add(name)
        np.testing.assert_array_equal(e, g, err_msg=f"case {case}: M={M} N={N} K={K} bias={bias is not None} relu={relu}")
    assert names == {"wq_gemm_i8_mfma_128", "wq_gemm_i8_mfma_256", "wq_gemm_i8_skinny"}


def test_random_shapes_f8_int4_against_the_general_kernel():
    """The same sweep for fp8 x int4 + MultiplyMultiply on the fp8 matrix core (both byte formats), on the exact input
    set: activations in {-2 .. 2}, power-of-two per-column and per-row factors."""
    rng = np.random.default_rng(650)
    lut_fnuz = {0: 0x00, 1: 0x40, 2: 0x48, -1: 0xC0, -2: 0xC8}   # E4M3 FNUZ (bias 8)
    lut_fn = {0: 0x00, 1: 0x38, 2: 0x40, -1: 0xB8, -2: 0xC0}     # OCP E4M3FN (bias 7)
    for case in range(24):
        M = int(rng.choice([65, 100, 256, 300, 512, 700]))
        N = int(rng.choice([256, 512, 1024]))
        K = int(rng.choice([128, 256, 640, 1024]))
        fn = bool(rng.integers(0, 2))
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        bp = pack_on_device(q, K, N)
        a_int = rng.integers(-2, 3, (M, K))
        a = np.vectorize((lut_fn if fn else lut_fnuz).get)(a_int).astype(np.uint8)
        d0 = (2.0 ** rng.integers(-3, 1, N)).astype(np.float32)
        d1 = (2.0 ** rng.integers(-3, 1, M)).astype(np.float32)
        kw = dict(a_dtype=capi.F8_E4M3FN if fn else capi.F8_FNUZ, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0, d1=d1)
        got = run_fpaintb(a, bp, None, M, N, K, flags=0 if fn else capi.FLAG_F8_NO_SPECIALS, expect_kernel=F8_MFMA, **kw)
        ref = run_fpaintb(a, bp, None, M, N, K, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC, **kw)
        np.testing.assert_array_equal(got, ref, err_msg=f"case {case}: M={M} N={N} K={K} e4m3fn={fn}")


def test_f8_int4_caller_split_k_in_128_deep_slices():
    """A caller's k_batch that does not divide K into whole 128-k tiles evenly (2048 / 3, / 5, / 7): the slices are cut in
    the fp8 kernel's tiles, trailing ones shorter or empty, and the sum equals the general kernel's single pass bit for
    bit on exact inputs."""
    rng = np.random.default_rng(651)
    lut_fn = {0: 0x00, 1: 0x38, 2: 0x40, -1: 0xB8, -2: 0xC0}     # OCP E4M3FN (bias 7)
    M, N, K = 300, 512, 2048
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    a = np.vectorize(lut_fn.get)(rng.integers(-2, 3, (M, K))).astype(np.uint8)
    d0 = (2.0 ** rng.integers(-6, -2, N)).astype(np.float32)
    d1 = (2.0 ** rng.integers(-3, 1, M)).astype(np.float32)
    kw = dict(a_dtype=capi.F8_E4M3FN, epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU, d0=d0, d1=d1)
    ref = run_fpaintb(a, bp, None, M, N, K, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC, **kw)
    for kb in (3, 5, 7, 17):
        got = run_fpaintb(a, bp, None, M, N, K, k_batch=kb, expect_kernel=F8_MFMA, **kw)
        np.testing.assert_array_equal(got, ref, err_msg=f"k_batch = {kb}")


@pytest.mark.parametrize("b_dtype", [capi.I4_PACKED, capi.I4_PRESHUFFLED])
def test_planner_split_k_on_the_256_tile_kernel(port, b_dtype):
    """M = 768 against 4096 columns: 48 tiles of 256 x 256, which the planner runs as five K slices on the 256-tile
    kernel when it gets a workspace and on the 128-tile kernel when it does not.  Exact inputs: both equal the general
    kernel bit for bit; realistic inputs: a row block of each against the CPU checker."""
    rng = np.random.default_rng(768)
    M, N, K = 768, 4096, 4096
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    packed = pack_on_device(q, K, N)
    b = packed if b_dtype == capi.I4_PACKED else preshuffle_on_device(packed, K, N)
    ai = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    si = f16_bits(rng.integers(1, 3, N).astype(np.float32))
    with_ws = run_fpaintb(ai, b, si, M, N, K, b_dtype=b_dtype, expect_kernel=MFMA256)
    without = run_fpaintb(ai, b, si, M, N, K, b_dtype=b_dtype, use_workspace=False)
    gen = run_fpaintb(ai, packed, si, M, N, K, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC)
    np.testing.assert_array_equal(with_ws, gen)
    np.testing.assert_array_equal(without, gen)
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    rows = slice(700, 732)
    exp = port.fpAintB_gemm_f16(np.ascontiguousarray(a[rows]), q, s, 32, N, K, K, K, N).reshape(32, N)
    for use_ws in (True, False):
        got = run_fpaintb(a, b, s, M, N, K, b_dtype=b_dtype, use_workspace=use_ws)
        check_err(bits_to_f32(got[rows], capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


def test_int8_requant_planner_split_k(port):
    """Short grids of the int8 x int8 GEMM (decode- and prefill-chunk-sized M): the planner splits K across blockIdx.z
    into int32 partials and requantises in the reduction.  Integer arithmetic: bit-equal to the CPU checker, to the
    general kernel and to the single-slice launch (no workspace), bias and Relu included; ragged M and N."""
    rng = np.random.default_rng(88)
    for (M, N, K, bias_on, relu) in ((16, 1024, 4096, False, False), (100, 516, 2048, True, True), (300, 1024, 8192, True, False),
                                     (1, 256, 1024, False, True), (64, 4096, 4096, True, True)):
        a = rng.integers(-128, 128, (M, K)).astype(np.int8)
        b = rng.integers(-128, 128, (N, K)).astype(np.int8)
        bias = rng.integers(-50000, 50001, N).astype(np.int32) if bias_on else None
        rq = 0.002
        e, name = _run_i8(a, b, M, N, K, rq, bias=bias, relu=relu)
        p = capi.fpAintB_problem(1 << 20, 1 << 20, None, 1 << 20, M, N, K, a_dtype=capi.I8, b_dtype=capi.I8, c_dtype=capi.I8,
                                 epilogue=capi.EPI_ACT_MUL_CLAMP, requant=rq)
        assert name == ("wq_gemm_i8_skinny" if M <= 16 else "wq_gemm_i8_mfma_128")
        if M > 1:
            assert capi.workspace_bytes(p) > 0 and "KBatch1>" not in capi.type_string(p), capi.type_string(p)
        one, _ = _run_i8(a, b, M, N, K, rq, bias=bias, relu=relu, use_workspace=False)
        g, gname = _run_i8(a, b, M, N, K, rq, bias=bias, relu=relu, flags=capi.FLAG_GENERIC)
        assert gname == GENERIC
        np.testing.assert_array_equal(e, g)
        np.testing.assert_array_equal(one, g)
        if bias is None and not relu:
            exp = port.gemm_i8_requant(a, b, M, N, K, K, K, N, rq).reshape(M, N)
            np.testing.assert_array_equal(e, exp)


def test_int8_requant_decode_on_the_streaming_kernel(port):
    """M <= 16 of the int8 x int8 GEMM: one pass over the weights on the streaming kernel, K dealt to the four waves of a
    workgroup (unevenly when it does not divide) and, for few columns, to several workgroups.  Integer arithmetic: bit-equal
    to the general kernel, to the 128-tile kernel and to the CPU checker; bias, Relu, padded rows of every operand, with and
    without the optional workspace."""
    rng = np.random.default_rng(1616)
    cases = [(1, 16, 256), (3, 48, 384), (16, 256, 640), (7, 1024, 1024), (16, 4096, 4096), (2, 11008, 4096), (16, 64, 8192),
             (5, 2048, 2304), (1, 512, 4608), (12, 4096, 1280), (16, 1024, 14336), (4, 128, 3072)]
    for i, (M, N, K) in enumerate(cases):
        a = rng.integers(-128, 128, (M, K + (32 if i % 3 == 1 else 0))).astype(np.int8)
        b = rng.integers(-128, 128, (N, K + (16 if i % 2 else 0))).astype(np.int8)
        bias = rng.integers(-50000, 50001, N).astype(np.int32) if i % 2 else None
        kw = dict(bias=bias, relu=bool(i % 3 == 0), lda=a.shape[1], ldb=b.shape[1], ldc=N + (12 if i % 4 == 2 else 0))
        rq = 0.0007
        e, name = _run_i8(a, b, M, N, K, rq, **kw)
        assert name == "wq_gemm_i8_skinny", (M, N, K, name)
        one, _ = _run_i8(a, b, M, N, K, rq, use_workspace=False, **kw)
        t, tname = _run_i8(a, b, M, N, K, rq, flags=capi.FLAG_TILE_128, **kw)
        g, gname = _run_i8(a, b, M, N, K, rq, flags=capi.FLAG_GENERIC, **kw)
        assert gname == GENERIC and tname == "wq_gemm_i8_mfma_128"
        for got in (e, one, t):
            np.testing.assert_array_equal(got[:, :N], g[:, :N], err_msg=f"M={M} N={N} K={K}")
        assert (e[:, N:] == 77).all()
        if bias is None and not kw["relu"] and a.shape[1] == K and b.shape[1] == K:
            np.testing.assert_array_equal(e[:, :N], port.gemm_i8_requant(a, b, M, N, K, K, K, N, rq).reshape(M, N))
    # saturating sums: every product 127 * 127 or -128 * 127
    M, N, K = 16, 256, 2048
    a = np.full((M, K), 127, np.int8)
    b = np.where(np.arange(N)[:, None] % 2 == 0, 127, -128).astype(np.int8) * np.ones((1, K), np.int8)
    e, name = _run_i8(a, b, M, N, K, 1e-5)
    g, _ = _run_i8(a, b, M, N, K, 1e-5, flags=capi.FLAG_GENERIC)
    assert name == "wq_gemm_i8_skinny"
    np.testing.assert_array_equal(e, g)


@pytest.mark.parametrize("fn", [True, False])
def test_f8_int4_decode_through_the_streaming_kernel(port, fn):
    """fp8 activations at M <= 16: the planner converts them to their (exact) fp16 image in the workspace and runs the
    small-M streaming kernel.  Every byte code of the format, the special ones included, against the general kernel
    bit for bit on exact data (all 256 codes appear; NaN positions compared as NaN), realistic data against the CPU
    checker; plain and pre-shuffled weights; without a workspace the fp8 tile kernel gives the same bits on exact data."""
    rng = np.random.default_rng(165)
    a_dtype = capi.F8_E4M3FN if fn else capi.F8_FNUZ
    lut = {0: 0x00, 1: 0x38, 2: 0x40, -1: 0xB8, -2: 0xC0} if fn else {0: 0x00, 1: 0x40, 2: 0x48, -1: 0xC0, -2: 0xC8}
    for (M, N, K, pre) in ((1, 2048, 1024, False), (8, 4096, 4096, True), (16, 2048, 2048, False)):
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        packed = pack_on_device(q, K, N)
        b = preshuffle_on_device(packed, K, N) if pre else packed
        b_dtype = capi.I4_PRESHUFFLED if pre else capi.I4_PACKED
        d0 = (2.0 ** rng.integers(-6, -2, N)).astype(np.float32)
        d1 = (2.0 ** rng.integers(-3, 1, M)).astype(np.float32)
        kw = dict(a_dtype=a_dtype, epilogue=capi.EPI_MULTIPLY_MULTIPLY_FASTGELU, d0=d0, d1=d1)
        a = np.vectorize(lut.get)(rng.integers(-2, 3, (M, K))).astype(np.uint8)
        # (the kernel widens the bytes itself while the fp16 image fits its LDS budget; else a first kernel writes the image)
        name = "wq_f8_fnuz_to_f16+wq_gemm_skinny_wsc" if M * (K * 2 + 16) > 40 * 1024 else "wq_gemm_skinny_ws"
        got = run_fpaintb(a, b, None, M, N, K, b_dtype=b_dtype, expect_kernel=name, **kw)
        ref = run_fpaintb(a, packed, None, M, N, K, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC, **kw)
        np.testing.assert_array_equal(got, ref)
        if fn:   # the fp8 tile kernel on the same problem: asked for by name, or reached by withholding the workspace
            tile = run_fpaintb(a, b, None, M, N, K, b_dtype=b_dtype, use_workspace=False,
                               flags=0 if pre else capi.FLAG_PINGPONG, **kw)
            np.testing.assert_array_equal(tile, ref)
        # every byte code, plain MultiplyMultiply: equal to the general kernel wherever that is a number, NaN where it is NaN
        a_all = rng.integers(0, 256, (M, K)).astype(np.uint8)
        kw2 = dict(a_dtype=a_dtype, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0, d1=d1)
        got = bits_to_f32(run_fpaintb(a_all, b, None, M, N, K, b_dtype=b_dtype, expect_kernel=name, **kw2), capi.F16)
        ref = bits_to_f32(run_fpaintb(a_all, packed, None, M, N, K, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC, **kw2), capi.F16)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        check_err(got[ok], ref[ok], 1e-3, 1e-3)


def test_torch_linear_preshuffled_weights(tmp_path):
    """Pre-shuffled int4 weights through the container and the torch operator: the same bits as the packed layout of
    the same weights (same kernel arithmetic), for a prefill-sized and a decode-sized batch and for an N that only the
    128-tile kernel takes."""
    from composable_kernel_amd import weights
    from composable_kernel_amd.torch_ops import WqLinear
    rng = np.random.default_rng(31)
    N, K = 512, 1024
    w = rng.normal(0, 0.05, (N, K)).astype(np.float32)
    q, s = weights.quantize_per_channel(w, 4, "f16")
    pk = weights.pack(q, s, 4, "f16")
    path = str(tmp_path / "pre.npz")
    weights.save_packed(path, weights.preshuffle(pk))
    pw = weights.load_packed(path)
    assert pw.layout == "preshuffled" and pw.b_dtype == capi.I4_PRESHUFFLED
    np.testing.assert_array_equal(weights.unshuffle(pw).weight, pk.weight)
    np.testing.assert_array_equal(weights.dequantize(pw), weights.dequantize(pk))
    lin_pre, lin_pk = WqLinear(pw), WqLinear(pk)
    x = torch.from_numpy(rng.uniform(-1, 1, (300, K)).astype(np.float32)).half().cuda()
    y = lin_pre(x)
    wd = torch.from_numpy(weights.dequantize(pk)).double()
    assert torch.allclose(y.cpu().double(), x.cpu().double() @ wd.T, rtol=2e-3, atol=2e-3)
    # the planner treats both layouts alike (same kernel, same split): equal bits for a prefill-sized and a decode-sized batch
    from composable_kernel_amd.torch_ops import fpaintb_linear
    for rows in (300, 5):
        assert torch.equal(lin_pre(x[:rows].contiguous()), lin_pk(x[:rows].contiguous()))
    # N = 384 is not a whole number of 256-column tiles: the 128-tile kernel reads the layout too
    q2, s2 = weights.quantize_per_channel(w[:384], 4, "f16")
    pk2 = weights.pack(q2, s2, 4, "f16")
    assert torch.equal(WqLinear(weights.preshuffle(pk2))(x), WqLinear(pk2)(x))
    with pytest.raises(ValueError):
        fpaintb_linear(x, lin_pre.weight, lin_pre.scale, 4, "none", None, "shuffled")
    # 8-bit weights, bf16 activations
    q8, s8 = weights.quantize_per_channel(w, 8, "bf16")
    pk8 = weights.pack(q8, s8, 8, "bf16")
    xb = x.to(torch.bfloat16)
    y8 = WqLinear(weights.preshuffle(pk8))(xb)
    wd8 = torch.from_numpy(weights.dequantize(pk8)).double()
    assert torch.allclose(y8.cpu().double(), xb.cpu().double() @ wd8.T, rtol=1.6e-2, atol=1.6e-2)
