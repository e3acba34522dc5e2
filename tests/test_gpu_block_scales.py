"""fp8 activations x int4 / int8 weights with block-wise scales of BOTH operands (the problem of
example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp with weight-quantized B), through the C ABI:
the general kernel bit-exact against outputs of the reference's own host check, the fp8 matrix-core kernel
(wq_gemm_f8_mfma_256_abscale: per-128-k block sums scaled by a_scale * b_scale) bit-exact on the exact set and within
check_err's tolerance of the checker on realistic data."""
import os

import numpy as np
import pytest
import torch

from composable_kernel_amd import capi
from gpu_util import bits_to_f32, check_err, dev, host_u16, pack_on_device

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "abscale_cases.npz")
TAGS = ["ab128_i4_bf16", "ab128_i4_f16_pow2", "ab_small_blocks_i8", "ab_ragged_i4"]
AB = "wq_gemm_f8_mfma_256_abscale"


def run(a, b_t, sa, sb, M, N, K, block, ld_sa, ld_sb, *, bf16, b_dtype, a_dtype=capi.F8_FNUZ, flags=0, expect=None,
        epilogue=capi.EPI_PASSTHROUGH, d0=None, d1=None, ldb=None, k_batch=1, workspace=True):
    a_t, sa_t, sb_t = dev(a), dev(np.ascontiguousarray(sa, np.float32)), dev(np.ascontiguousarray(sb, np.float32))
    c_t = torch.full((M + 16, N), 0x7BFF, dtype=torch.int16, device="cuda")
    keep = [dev(d0) if d0 is not None else None, dev(d1) if d1 is not None else None]
    p = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), None, c_t.data_ptr(), M, N, K, a_dtype=a_dtype, b_dtype=b_dtype,
                             c_dtype=capi.BF16 if bf16 else capi.F16, flags=flags, a_block_scale=sa_t.data_ptr(),
                             b_block_scale=sb_t.data_ptr(), scale_block=block, ld_a_scale=ld_sa, ld_b_scale=ld_sb, epilogue=epilogue,
                             d0=keep[0].data_ptr() if keep[0] is not None else None,
                             d1=keep[1].data_ptr() if keep[1] is not None else None, ldb=ldb, k_batch=k_batch)
    assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
    if expect:
        assert capi.kernel_name(p).startswith(expect), capi.kernel_name(p)
    nbytes = capi.workspace_bytes(p)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda") if nbytes and workspace else None
    capi.launch(p, ws.data_ptr() if ws is not None else None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = host_u16(c_t)
    assert (out[M:] == 0x7BFF).all()
    return out[:M]


@pytest.mark.parametrize("tag", TAGS)
def test_general_kernel_matches_the_reference_bit_for_bit(tag):
    g = np.load(GOLD)
    M, N, K, bm, bn, bk, ld_sa, ld_sb, bf = (int(v) for v in g[f"{tag}_shape"])
    q = g[f"{tag}_q"]
    exp = g[f"{tag}_e"].reshape(M, N)
    got = run(g[f"{tag}_a"], dev(q), g[f"{tag}_sa"], g[f"{tag}_sb"], M, N, K, (bm, bn, bk), ld_sa, ld_sb, bf16=bool(bf),
              b_dtype=capi.I8, flags=capi.FLAG_GENERIC, expect="wq_gemm_generic")
    np.testing.assert_array_equal(got, exp)
    if "i4" in tag and K % 8 == 0:   # the same weights packed two per byte
        got = run(g[f"{tag}_a"], pack_on_device(q, K, N), g[f"{tag}_sa"], g[f"{tag}_sb"], M, N, K, (bm, bn, bk), ld_sa, ld_sb,
                  bf16=bool(bf), b_dtype=capi.I4_PACKED, flags=capi.FLAG_GENERIC, expect="wq_gemm_generic")
        np.testing.assert_array_equal(got, exp)


def test_matrix_core_kernel_exact_set_golden():
    """Activations in {-2 .. 2}, power-of-two scales: every block sum and every scaled partial is exact, so the matrix-core
    kernel must give the reference's bits."""
    g = np.load(GOLD)
    tag = "ab128_i4_f16_pow2"
    M, N, K, bm, bn, bk, ld_sa, ld_sb, bf = (int(v) for v in g[f"{tag}_shape"])
    bp = pack_on_device(g[f"{tag}_q"], K, N)
    exp = g[f"{tag}_e"].reshape(M, N)
    got = run(g[f"{tag}_a"], bp, g[f"{tag}_sa"], g[f"{tag}_sb"], M, N, K, (bm, bn, bk), ld_sa, ld_sb, bf16=False,
              b_dtype=capi.I4_PACKED, flags=capi.FLAG_F8_NO_SPECIALS, expect=AB)
    np.testing.assert_array_equal(got, exp)
    pre = torch.empty_like(bp)
    capi.preshuffle_int4(bp.data_ptr(), K // 2, K, N, pre.data_ptr(), K, torch.cuda.current_stream().cuda_stream)
    got = run(g[f"{tag}_a"], pre, g[f"{tag}_sa"], g[f"{tag}_sb"], M, N, K, (bm, bn, bk), ld_sa, ld_sb, bf16=False,
              b_dtype=capi.I4_PRESHUFFLED, flags=capi.FLAG_F8_NO_SPECIALS, expect=AB)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("M,N,K,block,fn,bf", [(512, 512, 1024, (128, 128, 128), False, True), (256, 256, 128, (128, 128, 128), True, False),
                                               (300, 512, 256, (256, 32, 128), False, False), (768, 256, 384, (128, 64, 128), True, True),
                                               (1024, 1024, 4096, (128, 128, 128), False, True)])
def test_matrix_core_kernel_realistic(port, M, N, K, block, fn, bf):
    """Moderate-magnitude fp8 bytes, uniform scales in (0, 1): the matrix-core kernel against the checker within check_err's
    defaults for the output type (bf16 1e-2: one truncation step; fp16 1e-3), on 1, 2, 3 and many K tiles, a partial last
    row tile, padded scale tables, both fp8 encodings; and against the general kernel on the same launch arguments."""
    rng = np.random.default_rng(M + N + K)
    bm, bn, bk = block
    kb = K // bk
    a = (rng.integers(24, 88, (M, K)) + 128 * rng.integers(0, 2, (M, K))).astype(np.uint8)
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    ld_sa, ld_sb = kb + 1, kb + 2
    sa = rng.uniform(0.05, 1.0, ((M + bm - 1) // bm, ld_sa)).astype(np.float32)
    sb = rng.uniform(0.05, 1.0, ((N + bn - 1) // bn, ld_sb)).astype(np.float32)
    bp = pack_on_device(q, K, N)
    kw = dict(bf16=bf, b_dtype=capi.I4_PACKED, a_dtype=capi.F8_E4M3FN if fn else capi.F8_FNUZ)
    got = run(a, bp, sa, sb, M, N, K, block, ld_sa, ld_sb, flags=0 if fn else capi.FLAG_F8_NO_SPECIALS, expect=AB, **kw)
    gen = run(a, bp, sa, sb, M, N, K, block, ld_sa, ld_sb, flags=capi.FLAG_GENERIC, expect="wq_gemm_generic", **kw)
    dt = capi.BF16 if bf else capi.F16
    tol = 1e-2 if bf else 1e-3
    check_err(bits_to_f32(got, dt), bits_to_f32(gen, dt), tol, tol)
    rows = min(M, 24)
    exp = port.gemm_f8_i8_abscale(a[:rows], q, sa, sb, rows, N, K, K, K, N, block, ld_sa, ld_sb, e_bf16=bf, e4m3fn=fn).reshape(rows, N)
    np.testing.assert_array_equal(gen[:rows], exp)       # the general kernel is the reference's arithmetic
    check_err(bits_to_f32(got[:rows], dt), bits_to_f32(exp, dt), tol, tol)


def test_block_scales_with_an_epilogue_and_refusals(port):
    """A D-operand epilogue applies after the scaled sum (MultiplyMultiply here); malformed or unsupported combinations
    are refused with a message."""
    rng = np.random.default_rng(11)
    M, N, K, block = 256, 256, 256, (128, 128, 128)
    a = np.array([0xC8, 0xC0, 0x00, 0x40, 0x48], np.uint8)[rng.integers(0, 5, (M, K))]
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    sa = np.exp2(rng.integers(-2, 2, (2, 2))).astype(np.float32)
    sb = np.exp2(rng.integers(-2, 2, (2, 2))).astype(np.float32)
    d0 = np.exp2(rng.integers(-2, 1, N)).astype(np.float32)
    d1 = np.exp2(rng.integers(-2, 1, M)).astype(np.float32)
    bp = pack_on_device(q, K, N)
    base = run(a, bp, sa, sb, M, N, K, block, 2, 2, bf16=False, b_dtype=capi.I4_PACKED, flags=capi.FLAG_F8_NO_SPECIALS, expect=AB)
    got = run(a, bp, sa, sb, M, N, K, block, 2, 2, bf16=False, b_dtype=capi.I4_PACKED, flags=capi.FLAG_F8_NO_SPECIALS, expect=AB,
              epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0, d1=d1)
    exp = (bits_to_f32(base, capi.F16) * d0[None, :] * d1[:, None]).astype(np.float16)
    ok = np.abs(bits_to_f32(base, capi.F16)) < 1024      # (where the unscaled result is itself exact in fp16)
    np.testing.assert_array_equal(got[ok], exp.view(np.uint16)[ok])
    a_t, sa_t, sb_t = dev(a), dev(sa), dev(sb)
    c_t = torch.zeros((M, N), dtype=torch.int16, device="cuda")

    def prob(**kw):
        args = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, a_block_scale=sa_t.data_ptr(), b_block_scale=sb_t.data_ptr(),
                    scale_block=block)
        args.update(kw)
        return capi.fpAintB_problem(a_t.data_ptr(), bp.data_ptr(), None, c_t.data_ptr(), M, N, K, **args)

    assert capi.is_supported(prob())
    assert not capi.is_supported(prob(b_block_scale=None))
    assert b"pairs" in capi.lib.wqgemm_last_error()
    assert not capi.is_supported(prob(scale_block=(128, 128, 0)))
    assert not capi.is_supported(prob(ld_a_scale=1))
    half = dev(np.zeros((M, K), np.uint16))
    p16 = capi.fpAintB_problem(half.data_ptr(), bp.data_ptr(), None, c_t.data_ptr(), M, N, K, a_block_scale=sa_t.data_ptr(),
                               b_block_scale=sb_t.data_ptr(), scale_block=block)
    assert not capi.is_supported(p16)
    # block sizes the matrix-core kernel does not take run on the general kernel
    assert capi.kernel_name(prob(scale_block=(64, 128, 128))) == "wq_gemm_f8_mfma_256_abscale_rows"   # finer activation blocks
    assert capi.kernel_name(prob(scale_block=(128, 16, 128))) == "wq_gemm_generic"
    assert capi.kernel_name(prob(scale_block=(128, 128, 64))) == "wq_gemm_generic"


@pytest.mark.parametrize("M,N,K,kb,fn", [(512, 512, 2048, 2, True), (256, 256, 1536, 3, False), (300, 512, 1024, 4, True)])
def test_block_scales_with_split_k(M, N, K, kb, fn):
    """k_batch slices of a block-scaled problem: the scale tables are indexed with the absolute k and the fp32 partials carry
    the scales, so on data whose partial sums are exact (small integers, power-of-two scales) the split result equals the
    single pass bit for bit -- on the matrix-core kernel and on the general kernel."""
    rng = np.random.default_rng(K + kb)
    block = (128, 128, 128)
    one = 0x38 if fn else 0x40
    lut = np.array([0x80 | (one + 8), 0x80 | one, 0, one, one + 8], np.uint8)
    a = lut[rng.integers(0, 5, (M, K))]
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    kblocks = K // 128
    sa = np.exp2(rng.integers(-4, 0, ((M + 127) // 128, kblocks))).astype(np.float32)
    sb = np.exp2(rng.integers(-4, 0, (N // 128, kblocks))).astype(np.float32)
    bp = pack_on_device(q, K, N)
    kw = dict(bf16=False, b_dtype=capi.I4_PACKED, a_dtype=capi.F8_E4M3FN if fn else capi.F8_FNUZ)
    flags = 0 if fn else capi.FLAG_F8_NO_SPECIALS
    single = run(a, bp, sa, sb, M, N, K, block, kblocks, kblocks, flags=flags, expect=AB, **kw)
    split = run(a, bp, sa, sb, M, N, K, block, kblocks, kblocks, flags=flags, expect=AB, k_batch=kb, **kw)
    np.testing.assert_array_equal(split, single)
    gen = run(a, bp, sa, sb, M, N, K, block, kblocks, kblocks, flags=capi.FLAG_GENERIC, expect="wq_gemm_generic", k_batch=kb, **kw)
    np.testing.assert_array_equal(gen, single)


def test_block_scales_planner_split_for_a_short_grid():
    """A grid far shorter than the chip (64 rows against 1024 columns, K = 4096) gets the planner's own K split when a
    workspace is there (kernel name with the reduce stage), one slice without it; same bits either way on exact data."""
    rng = np.random.default_rng(5)
    M, N, K, block = 64, 1024, 4096, (128, 128, 128)
    lut = np.array([0xC0, 0xB8, 0, 0x38, 0x40], np.uint8)
    a = lut[rng.integers(0, 5, (M, K))]
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    sa = np.exp2(rng.integers(-4, 0, (1, K // 128))).astype(np.float32)
    sb = np.exp2(rng.integers(-4, 0, (N // 128, K // 128))).astype(np.float32)
    bp = pack_on_device(q, K, N)
    kw = dict(bf16=True, b_dtype=capi.I4_PACKED, a_dtype=capi.F8_E4M3FN)
    with_ws = run(a, bp, sa, sb, M, N, K, block, 0, 0, expect=AB, **kw)
    without = run(a, bp, sa, sb, M, N, K, block, 0, 0, expect=AB, workspace=False, **kw)
    np.testing.assert_array_equal(with_ws, without)
    gen = run(a, bp, sa, sb, M, N, K, block, 0, 0, flags=capi.FLAG_GENERIC, expect="wq_gemm_generic", **kw)
    np.testing.assert_array_equal(with_ws, gen)


@pytest.mark.parametrize("M,N,K,bm,bn,fn,bf,kb", [(512, 512, 1024, 1, 128, True, True, 1), (300, 256, 384, 1, 128, False, False, 1),
                                                  (256, 768, 128, 32, 64, True, False, 1), (640, 256, 2048, 100, 128, False, True, 1),
                                                  (512, 512, 2048, 1, 128, True, True, 2), (1024, 1024, 4096, 1, 128, True, True, 1)])
def test_per_row_activation_scales_on_the_matrix_core(port, M, N, K, bm, bn, fn, bf, kb):
    """Activation scales finer than a 128-row tile half -- one per token (scale_block_m = 1, the usual serving recipe) or per
    any other row block -- run on wq_gemm_f8_mfma_256_abscale_rows: the row scales of a K tile travel with it through LDS.
    Exact data (small integers, power-of-two scales): bit-equal to the general kernel, which has the reference's arithmetic;
    realistic data: within check_err's defaults of the checker."""
    rng = np.random.default_rng(M + K + bm)
    block = (bm, bn, 128)
    kblocks = K // 128
    ld_sa, ld_sb = kblocks + 1, kblocks
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    bp = pack_on_device(q, K, N)
    kw = dict(bf16=bf, b_dtype=capi.I4_PACKED, a_dtype=capi.F8_E4M3FN if fn else capi.F8_FNUZ)
    flags = 0 if fn else capi.FLAG_F8_NO_SPECIALS
    rows_k = "wq_gemm_f8_mfma_256_abscale_rows"
    # exact set
    one = 0x38 if fn else 0x40
    lut = np.array([0x80 | (one + 8), 0x80 | one, 0, one, one + 8], np.uint8)
    a = lut[rng.integers(0, 5, (M, K))]
    sa = np.exp2(rng.integers(-4, 1, ((M + bm - 1) // bm, ld_sa))).astype(np.float32)
    sb = np.exp2(rng.integers(-4, 1, ((N + bn - 1) // bn, ld_sb))).astype(np.float32)
    got = run(a, bp, sa, sb, M, N, K, block, ld_sa, ld_sb, flags=flags, expect=rows_k, k_batch=kb, **kw)
    gen = run(a, bp, sa, sb, M, N, K, block, ld_sa, ld_sb, flags=capi.FLAG_GENERIC, expect="wq_gemm_generic", **kw)
    np.testing.assert_array_equal(got, gen)
    # realistic set
    a = (rng.integers(24, 88, (M, K)) + 128 * rng.integers(0, 2, (M, K))).astype(np.uint8)
    sa = rng.uniform(0.05, 1.0, sa.shape).astype(np.float32)
    sb = rng.uniform(0.05, 1.0, sb.shape).astype(np.float32)
    got = run(a, bp, sa, sb, M, N, K, block, ld_sa, ld_sb, flags=flags, expect=rows_k, k_batch=kb, **kw)
    rows = min(M, 20)
    exp = port.gemm_f8_i8_abscale(a[:rows], q, sa, sb, rows, N, K, K, K, N, block, ld_sa, ld_sb, e_bf16=bf, e4m3fn=fn).reshape(rows, N)
    dt = capi.BF16 if bf else capi.F16
    tol = 1e-2 if bf else 1e-3
    check_err(bits_to_f32(got[:rows], dt), bits_to_f32(exp, dt), tol, tol)
    tail = run(a[M - 8:], bp, sa[(M - 8) // bm:], sb, 8, N, K, block, ld_sa, ld_sb, flags=capi.FLAG_GENERIC, expect="wq_gemm_generic", **kw) \
        if (M - 8) % bm == 0 else None
    if tail is not None:   # the last rows too (a partial last row tile for M = 300)
        check_err(bits_to_f32(got[M - 8:], dt), bits_to_f32(tail, dt), tol, tol)
