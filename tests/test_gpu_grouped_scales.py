"""Group-wise weight scales (one scale of the activation type per column and per group of g consecutive k) on the GPU, through the C ABI:
the committed goldens (produced by the reference's ReferencefpAintBGemm on the (K, N) scale tensor whose rows repeat per
group) and the CPU checker at larger sizes.

Bars: bit-exact on the general kernel (fp32 accumulation in the reference's ascending-k order) and on every kernel for
integer-valued inputs; ck::utils::check_err's half defaults (rtol 1e-3, atol 1e-3 --
library/include/ck/library/utility/check_err.hpp:82-86) for real-valued inputs on the matrix-core kernels."""
import numpy as np
import pytest

from composable_kernel_amd import capi
from gpu_util import bits_to_f32, check_err, f16_bits, pack_on_device, run_fpaintb

pytestmark = pytest.mark.gpu

MFMA, MFMA256, GENERIC = "wq_gemm_mfma_128", "wq_gemm_mfma_256", "wq_gemm_generic"


def _weights(q, K, N, b_dtype):
    if b_dtype == capi.I4_PACKED:
        return pack_on_device(q, K, N)
    return q if b_dtype == capi.I8 else (q.astype(np.int16) + 128).astype(np.uint8)


@pytest.mark.parametrize("tag,b_dtype,tiled", [("grp64_i4", capi.I4_PACKED, True), ("grp128_i4", capi.I4_PACKED, True),
                                               ("grp64_i8", capi.I8, True), ("grp_ragged_i4", capi.I4_PACKED, False),
                                               ("grp128_int_i4", capi.I4_PACKED, True)])
def test_grouped_goldens(golden_gemm, tag, b_dtype, tiled):
    g = golden_gemm
    M, N, K, lda, ldb, ldc, group_k, ld_scale = (int(v) for v in g[f"{tag}_shape"])
    q = np.ascontiguousarray(g[f"{tag}_b"][:, :K])
    b = _weights(q, K, N, b_dtype)
    exp = np.zeros(M * ldc, np.uint16)
    exp[: g[f"{tag}_c"].size] = g[f"{tag}_c"]
    exp = exp.reshape(M, ldc)[:, :N]
    got = run_fpaintb(g[f"{tag}_a"], b, g[f"{tag}_s"], M, N, K, b_dtype=b_dtype, lda=lda, ldc=ldc, flags=capi.FLAG_GENERIC,
                      scale_group_k=group_k, ld_scale=ld_scale, expect_kernel=GENERIC)
    np.testing.assert_array_equal(got[:, :N], exp)
    # default route: the small-M kernel for the golden cases of up to 64 rows, else the 128-tile kernel
    want = ("wq_gemm_skinny" if M <= 64 and N % 64 == 0 else MFMA) if tiled else GENERIC
    got = run_fpaintb(g[f"{tag}_a"], b, g[f"{tag}_s"], M, N, K, b_dtype=b_dtype, lda=lda, ldc=ldc,
                      scale_group_k=group_k, ld_scale=ld_scale, expect_kernel=want)
    if tiled and want != MFMA:   # and the 128-tile kernel by name
        got128 = run_fpaintb(g[f"{tag}_a"], b, g[f"{tag}_s"], M, N, K, b_dtype=b_dtype, lda=lda, ldc=ldc, flags=capi.FLAG_TILE_128,
                             scale_group_k=group_k, ld_scale=ld_scale, expect_kernel=MFMA)
        check_err(bits_to_f32(got128[:, :N], capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
    if tag == "grp128_int_i4" or not tiled:
        np.testing.assert_array_equal(got[:, :N], exp)     # integer-valued inputs (planner-chosen split-K on top)
    else:
        check_err(bits_to_f32(got[:, :N], capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


@pytest.mark.parametrize("b_dtype", [capi.I4_PACKED, capi.I8, capi.U8_OFFSET])
@pytest.mark.parametrize("group_k", [64, 128, 256])
@pytest.mark.parametrize("flags,kernel", [(capi.FLAG_TILE_128, MFMA), (capi.FLAG_TILE_256, MFMA256)])
def test_grouped_tile_kernels_vs_checker(port, b_dtype, group_k, flags, kernel):
    rng = np.random.default_rng(group_k + b_dtype)
    M, N, K = 300, 512, 768                      # partial last row tile; three / six / twelve scale groups
    lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
    q = rng.integers(lo, hi, (N, K)).astype(np.int8)
    b = _weights(q, K, N, b_dtype)
    groups, ld_scale = K // group_k, N + 8
    # integer-valued: exact in any accumulation order
    a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = f16_bits(rng.integers(1, 4, (groups, ld_scale)).astype(np.float32))
    got = run_fpaintb(a, b, s, M, N, K, b_dtype=b_dtype, flags=flags, scale_group_k=group_k, ld_scale=ld_scale,
                      expect_kernel=kernel)
    exp = port.fpAintB_gemm_f16_grouped(a, q, s, M, N, K, K, K, N, group_k, ld_scale).reshape(M, N)
    np.testing.assert_array_equal(got, exp)
    # real-valued
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -10, 2.0 ** -7, (groups, ld_scale)) * (16 if b_dtype == capi.I4_PACKED else 1))
    got = run_fpaintb(a, b, s, M, N, K, b_dtype=b_dtype, flags=flags, scale_group_k=group_k, ld_scale=ld_scale,
                      expect_kernel=kernel)
    exp = port.fpAintB_gemm_f16_grouped(a, q, s, M, N, K, K, K, N, group_k, ld_scale).reshape(M, N)
    check_err(bits_to_f32(got, capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)


@pytest.mark.parametrize("k_batch", [2, 3])
@pytest.mark.parametrize("flags,kernel", [(capi.FLAG_TILE_128, MFMA), (capi.FLAG_TILE_256, MFMA256), (capi.FLAG_GENERIC, GENERIC)])
def test_grouped_with_split_k(port, k_batch, flags, kernel):
    """K slices start in the middle of the K range: the group index follows the absolute k."""
    rng = np.random.default_rng(k_batch)
    M, N, K, group_k = 256, 256, 640, 128
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    a = f16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = f16_bits(rng.integers(1, 4, (K // group_k, N)).astype(np.float32))
    got = run_fpaintb(a, pack_on_device(q, K, N), s, M, N, K, flags=flags, k_batch=k_batch, scale_group_k=group_k,
                      expect_kernel=kernel)
    exp = port.fpAintB_gemm_f16_grouped(a, q, s, M, N, K, K, K, N, group_k, N).reshape(M, N)
    np.testing.assert_array_equal(got, exp)


def test_grouped_default_route_full_width(port):
    """A grid that fills the chip takes the 256-tile kernel by itself; rows checked against the CPU checker."""
    rng = np.random.default_rng(77)
    M, N, K, group_k = 2048, 4096, 512, 128
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    a = f16_bits(rng.uniform(-1, 1, (M, K)))
    s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, (K // group_k, N)))
    got = run_fpaintb(a, pack_on_device(q, K, N), s, M, N, K, scale_group_k=group_k, expect_kernel=MFMA256)
    for r0 in (0, 1984):
        rows = np.ascontiguousarray(a[r0:r0 + 64])
        exp = port.fpAintB_gemm_f16_grouped(rows, q, s, 64, N, K, K, K, N, group_k, N).reshape(64, N)
        check_err(bits_to_f32(got[r0:r0 + 64], capi.F16), bits_to_f32(exp, capi.F16), 1e-3, 1e-3)
    # per-channel scales are the one-group case
    s1 = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, (1, N)))
    one_group = run_fpaintb(a, pack_on_device(q, K, N), s1, M, N, K, scale_group_k=K, expect_kernel=MFMA256)
    per_channel = run_fpaintb(a, pack_on_device(q, K, N), s1[0], M, N, K, expect_kernel=MFMA256)
    np.testing.assert_array_equal(one_group, per_channel)


# ---- bf16 activations: Multiply-on-B with a B1 tensor whose rows repeat per group (the checker is pinned by the reference's
# Multiply functor + ReferenceGemm + epilogue functors; goldens `bgrp*`).  Bars: bit-exact on the general kernel and, for
# integer-valued inputs, on every kernel; rtol 1e-2 / atol 1e-3 (one bf16 truncation step) for real-valued inputs on the
# matrix-core kernels, whose fp32 summation order differs.
from gpu_util import bf16_bits  # noqa: E402

_EPI = {0: capi.EPI_PASSTHROUGH, 1: capi.EPI_FASTGELU, 3: capi.EPI_ADD, 8: capi.EPI_CONVERT_BF16_RTN}


@pytest.mark.parametrize("tag,b_dtype", [("bgrp64_i8", capi.I8), ("bgrp128_i4", capi.I4_PACKED), ("bgrp32_i8_add", capi.I8),
                                         ("bgrp64_i4_rtn", capi.I4_PACKED), ("bgrp128_int_i4", capi.I4_PACKED)])
def test_grouped_goldens_bf16(golden_gemm, tag, b_dtype):
    g = golden_gemm
    M, N, K, group_k, ld_b1, ep = (int(v) for v in g[f"{tag}_shape"])
    b = _weights(g[f"{tag}_b"], K, N, b_dtype)
    exp = g[f"{tag}_e"].reshape(M, N)
    kw = dict(a_dtype=capi.BF16, b_dtype=b_dtype, epilogue=_EPI[ep], d0=g[f"{tag}_d0"] if ep == 3 else None,
              scale_group_k=group_k, ld_scale=ld_b1)
    got = run_fpaintb(g[f"{tag}_a"], b, g[f"{tag}_s"], M, N, K, flags=capi.FLAG_GENERIC, expect_kernel=GENERIC, **kw)
    if ep == 1:   # FastGelu: device exponential against the host's, one bf16 step
        check_err(bits_to_f32(got, capi.BF16), bits_to_f32(exp, capi.BF16), 1e-2, 1e-3)
    else:
        np.testing.assert_array_equal(got, exp)
    if tag == "bgrp128_int_i4":
        for flags, kernel in ((capi.FLAG_TILE_128, MFMA),):
            got = run_fpaintb(g[f"{tag}_a"], b, g[f"{tag}_s"], M, N, K, flags=flags, expect_kernel=kernel, **kw)
            np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("b_dtype", [capi.I4_PACKED, capi.I8])
@pytest.mark.parametrize("group_k", [64, 128])
@pytest.mark.parametrize("flags,kernel", [(capi.FLAG_TILE_128, MFMA), (capi.FLAG_TILE_256, MFMA256)])
def test_grouped_tile_kernels_vs_checker_bf16(port, b_dtype, group_k, flags, kernel):
    rng = np.random.default_rng(group_k + b_dtype + 100)
    M, N, K = 300, 512, 768
    lo, hi = (-8, 8) if b_dtype == capi.I4_PACKED else (-128, 128)
    q = rng.integers(lo, hi, (N, K)).astype(np.int8)
    b = _weights(q, K, N, b_dtype)
    groups, ld_scale = K // group_k, N + 8
    kw = dict(a_dtype=capi.BF16, b_dtype=b_dtype, flags=flags, scale_group_k=group_k, ld_scale=ld_scale, expect_kernel=kernel)
    a = bf16_bits(np.round(rng.uniform(-2, 2, (M, K))))
    s = bf16_bits(rng.integers(1, 4, (groups, ld_scale)).astype(np.float32))
    got = run_fpaintb(a, b, s, M, N, K, **kw)
    exp = port.gemm_bf16_i8_grouped(a, q, s, M, N, K, K, 1, K, N, 0, group_k, ld_scale).reshape(M, N)
    np.testing.assert_array_equal(got, exp)
    a = bf16_bits(rng.uniform(-1, 1, (M, K)))
    s = bf16_bits(rng.uniform(2.0 ** -10, 2.0 ** -7, (groups, ld_scale)) * (16 if b_dtype == capi.I4_PACKED else 1))
    got = run_fpaintb(a, b, s, M, N, K, **kw)
    exp = port.gemm_bf16_i8_grouped(a, q, s, M, N, K, K, 1, K, N, 0, group_k, ld_scale).reshape(M, N)
    check_err(bits_to_f32(got, capi.BF16), bits_to_f32(exp, capi.BF16), 1e-2, 1e-3)


@pytest.mark.parametrize("M,N,K,g", [(1, 2048, 4096, 128), (8, 2048, 512, 64), (24, 128, 256, 32), (40, 256, 1024, 128),
                                     (64, 512, 2048, 128)])
def test_grouped_small_m_kernels_bf16(port, M, N, K, g):
    rng = np.random.default_rng(M + g)
    for b_dtype, (lo, hi) in ((capi.I4_PACKED, (-8, 8)), (capi.I8, (-128, 128))):
        name = capi.kernel_name(capi.fpAintB_problem(1 << 20, 1 << 20, 1 << 20, 1 << 20, M, N, K, a_dtype=capi.BF16, b_dtype=b_dtype,
                                                     scale_group_k=g))
        assert name.startswith("wq_gemm_skinny"), name
        q = rng.integers(lo, hi, (N, K)).astype(np.int8)
        b = _weights(q, K, N, b_dtype)
        a = bf16_bits(np.round(rng.uniform(-2, 2, (M, K))))
        s = bf16_bits(rng.integers(1, 4, (K // g, N)).astype(np.float32))
        got = run_fpaintb(a, b, s, M, N, K, a_dtype=capi.BF16, b_dtype=b_dtype, scale_group_k=g, expect_kernel=name)
        exp = port.gemm_bf16_i8_grouped(a, q, s, M, N, K, K, 1, K, N, 0, g, N).reshape(M, N)
        np.testing.assert_array_equal(got, exp)
