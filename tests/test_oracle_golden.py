"""The plain-C checker (oracle/wq_oracle.c) against golden vectors produced by the reference's
own host code (tests/golden/make_golden.py).  Bit-exact everywhere: these are integer / bit-level
results or fp32 loops with a fixed evaluation order."""
import os

import numpy as np
import pytest


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def test_dequant_f16_table(port, golden_scalars):
    g = golden_scalars
    q = np.arange(-128, 128)
    for s, row in zip(g["dequant_f16_scales"], g["dequant_f16"]):
        got = np.array([port.dequant_f16(int(v), int(s)) for v in q], np.uint16)
        np.testing.assert_array_equal(got, row)


def test_multiply_bf16_table(port, golden_scalars):
    g = golden_scalars
    q = np.arange(-128, 128)
    for s, row in zip(g["multiply_bf16_scales"], g["multiply_bf16"]):
        got = np.array([port.multiply_bf16(int(v), int(s)) for v in q], np.uint16)
        np.testing.assert_array_equal(got, row)


def test_conversions(port, golden_scalars):
    g = golden_scalars
    fx = g["f32_in"]
    np.testing.assert_array_equal(np.array([port.f32_to_f16(float(v)) for v in fx], np.uint16), g["f32_to_f16"])
    np.testing.assert_array_equal(np.array([port.f32_to_bf16(float(v)) for v in fx], np.uint16), g["f32_to_bf16"])
    got = np.array([port.f16_to_f32(h) for h in range(0, 65536, 7)], np.float32)
    exp = g["f16_to_f32"]
    nan = np.isnan(exp)
    assert np.array_equal(np.isnan(got), nan)
    np.testing.assert_array_equal(_bits(got[~nan]), _bits(exp[~nan]))


def test_f8_fnuz_table(port, golden_scalars):
    exp = golden_scalars["f8_to_f32"]
    got = np.array([port.f8_to_f32(v) for v in range(256)], np.float32)
    assert np.isnan(got[0x80]) and np.isnan(exp[0x80])
    m = np.arange(256) != 0x80
    np.testing.assert_array_equal(_bits(got[m]), _bits(exp[m]))
    # known answers held by the reference's own test (test/data_type/test_fp8.cpp:13-54)
    assert got[0x08] == 2.0 ** -7 and got[0x7F] == 240.0 and got[0xFF] == -240.0
    assert got[0x01] == 2.0 ** -10 and got[0x40] == 1.0 and got[0xC0] == -1.0


def test_fastgelu(port, golden_scalars):
    g = golden_scalars
    got = np.array([port.fastgelu(float(v)) for v in g["fastgelu_in"]], np.float32)
    np.testing.assert_array_equal(_bits(got), _bits(g["fastgelu"]))


def test_activation_mul_clamp(port, golden_scalars):
    g = golden_scalars
    acc = g["amc_in"]
    np.testing.assert_array_equal(np.array([port.act_mul_clamp(0.03, int(v)) for v in acc], np.int8), g["amc_0p03"])
    np.testing.assert_array_equal(
        np.array([port.act_mul_clamp_from_float(0.03, float(v)) for v in acc], np.int8), g["amc_from_float_0p03"])


def test_multiply_multiply_and_mafg(port, golden_scalars):
    g = golden_scalars
    c, d0, d1 = g["mm_c"], g["mm_d0"], g["mm_d1"]
    np.testing.assert_array_equal(
        np.array([port.multiply_multiply_f16(float(a), float(b), float(e)) for a, b, e in zip(c, d0, d1)], np.uint16),
        g["mm_f16"])
    np.testing.assert_array_equal(
        np.array([port.multiply_multiply_bf16(float(a), float(b), float(e)) for a, b, e in zip(c, d0, d1)], np.uint16),
        g["mm_bf16"])
    np.testing.assert_array_equal(
        np.array([port.multiply_add_fastgelu_bf16(float(a), int(x), int(y))
                  for a, x, y in zip(c, g["mafg_d0"], g["mafg_d1"])], np.uint16),
        g["mafg_bf16"])


def test_generators(port, golden_scalars):
    g = golden_scalars
    for dt in ("f16", "bf16", "i8", "f32"):
        for iv in (0, 1):
            got = port.fill_uniform(96, dt, -1.0, 1.0, bool(iv))
            exp = g[f"fill_{dt}_{iv}"]
            if dt == "f32":
                np.testing.assert_array_equal(_bits(got), _bits(exp))
            else:
                np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(port.fill_uniform(96, "i8", -8.0, 7.0, True), g["fill_i8_int4range"])
    np.testing.assert_array_equal(port.generator2_i8(8, 12, 12, 1, -128, 127, seed=1), g["gen2_i8_seed1"])
    np.testing.assert_array_equal(port.generator2_i8(8, 12, 1, 8, -5, 5, seed=1), g["gen2_i8_seed1_col"])


def test_fpaintb_cases(port, golden_gemm):
    g = golden_gemm
    for tag in ("fpaintb_real", "fpaintb_int", "fpaintb_ragged", "fpaintb_i8"):
        M, N, K, lda, ldb, ldc = (int(v) for v in g[f"{tag}_shape"])
        got = port.fpAintB_gemm_f16(g[f"{tag}_a"], g[f"{tag}_b"], g[f"{tag}_s"], M, N, K, lda, ldb, ldc)
        np.testing.assert_array_equal(got, g[f"{tag}_c"], err_msg=tag)


def test_example64_inputs_and_outputs(port, golden_gemm):
    # example/64 fills A, quant-B and scale each from a fresh mt19937(11939)
    for init in (1, 2):
        M, N, K = 32, 32, 64
        a = port.fill_uniform(M * K, "f16", -1.0, 1.0, init == 1).reshape(M, K)
        b = port.fill_uniform(K * N, "i8", -1.0, 1.0, init == 1).reshape(N, K)
        s = port.fill_uniform(N, "f16", -1.0, 1.0, init == 1)
        got = port.fpAintB_gemm_f16(a, b, s, M, N, K, K, K, N)
        np.testing.assert_array_equal(got, golden_gemm[f"ex64_init{init}_c"])


def test_c1_int8_requant_256(port, golden_gemm):
    """BASELINE config 1: int8 x int8 M=N=K=256 through the host ReferenceGemm of example/14."""
    M = N = K = 256
    a = port.generator2_i8(M, K, K, 1, -128, 127, seed=1)
    b = port.generator2_i8(K, N, 1, K, -128, 127, seed=-1)
    np.testing.assert_array_equal(a[:64], golden_gemm["c1_a_head"])
    np.testing.assert_array_equal(b[:64], golden_gemm["c1_b_head"])
    e = port.gemm_i8_requant(a, b, M, N, K, K, K, N, 0.03).reshape(M, N)
    np.testing.assert_array_equal(e, golden_gemm["c1_e"])


def test_i8_small_ragged(port, golden_gemm):
    g = golden_gemm
    e = port.gemm_i8_requant(g["i8_small_a"], g["i8_small_b"], 40, 24, 96, 104, 100, 32, 0.25)
    np.testing.assert_array_equal(e, g["i8_small_e"])


def test_i8_bias_relu_requant(port, golden_gemm):
    """example/14's bias + relu verify path: srand(1) inputs (A, B, then the int32 bias) at 128^3, and the small
    ragged case, for Activation in {PassThrough, Relu} with and without the bias."""
    g = golden_gemm
    M = N = K = 128
    a = port.generator2_i8(M, K, K, 1, -128, 127, seed=1)
    b = port.generator2_i8(K, N, 1, K, -128, 127, seed=-1)
    bias = port.generator2_i8(1, N, N, 1, -128, 127, seed=-1).astype(np.int32)
    np.testing.assert_array_equal(bias, g["i8b_bias"])
    for relu in (0, 1):
        np.testing.assert_array_equal(port.gemm_i8_bias_requant(a, b, None, M, N, K, K, K, N, 0.03, relu),
                                      g[f"i8b_e_relu{relu}"])
        np.testing.assert_array_equal(port.gemm_i8_bias_requant(a, b, bias, M, N, K, K, K, N, 0.03, relu),
                                      g[f"i8b_e_bias_relu{relu}"])
        np.testing.assert_array_equal(
            port.gemm_i8_bias_requant(g["i8_small_a"], g["i8_small_b"], g["i8_small_bias"], 40, 24, 96, 104, 100, 32,
                                      0.25, relu), g[f"i8_small_e_bias_relu{relu}"])
        np.testing.assert_array_equal(
            port.gemm_i8_bias_requant(g["i8_small_a"], g["i8_small_b"], None, 40, 24, 96, 104, 100, 32, 0.25, relu),
            g[f"i8_small_e_relu{relu}"])
    # without Relu and bias, and while the float accumulation of the plain path is exact, both paths agree
    np.testing.assert_array_equal(g["i8_small_e_relu0"], g["i8_small_e"])


def test_bf16_i8_cases(port, golden_gemm):
    g = golden_gemm
    M, N, K = 48, 64, 128
    a, s, bcol, d0, d1 = g["bf16_a"], g["bf16_s"], g["bf16_bcol"], g["bf16_d0"], g["bf16_d1"]
    for ep in (0, 1, 2):
        got = port.gemm_bf16_i8(a, bcol, s, M, N, K, K, 1, K, N, ep, d0, d1)
        np.testing.assert_array_equal(got, g[f"bf16_e_col_ep{ep}"], err_msg=f"epilogue {ep}")
    brow = np.ascontiguousarray(bcol.T)
    got = port.gemm_bf16_i8(a, brow, s, M, N, K, K, N, 1, N, 1, d0, d1)
    np.testing.assert_array_equal(got, g["bf16_e_row_ep1"])
    # binary epilogues of the multi-ABD instance family, and the "multiply" instances (no Multiply on B)
    for ep in (3, 4, 5, 6):
        got = port.gemm_bf16_i8(a, bcol, s, M, N, K, K, 1, K, N, ep, d0, d1)
        np.testing.assert_array_equal(got, g[f"bf16_e_col_ep{ep}"], err_msg=f"epilogue {ep}")
    for ep in (0, 2, 5, 6):
        got = port.gemm_bf16_i8(a, brow, None, M, N, K, K, N, 1, N, ep, s, d1)
        np.testing.assert_array_equal(got, g[f"bf16_mul_e_row_ep{ep}"], err_msg=f"multiply instance, epilogue {ep}")


def test_f8_i4_multiply_multiply(port, golden_gemm):
    g = golden_gemm
    M, N, K = 48, 64, 128
    for key, gelu in (("f8_e_mm", False), ("f8_e_mm_gelu", True)):
        got = port.gemm_f8_i8_mm(g["f8_a"], g["f8_b"], g["f8_d0"], g["f8_d1"], M, N, K, K, K, N, gelu)
        np.testing.assert_array_equal(got, g[key], err_msg=key)


def test_pack_int4_roundtrip_and_layout(port):
    rng = np.random.default_rng(1)
    K, N = 64, 5
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    p = port.pack_int4(q, K, N)
    assert p.shape == (N, K // 2)
    np.testing.assert_array_equal(port.unpack_int4(p, K, N), q)
    # layout: 32-bit word g of column n, nibble p<4 -> k=8g+2p, p>=4 -> k=8g+2(p-4)+1, value q+8
    w = p.reshape(N, K // 8, 4).astype(np.uint32)
    w = w[..., 0] | (w[..., 1] << 8) | (w[..., 2] << 16) | (w[..., 3] << 24)
    for pos in range(8):
        k_in = 2 * pos if pos < 4 else 2 * (pos - 4) + 1
        nib = (w >> (4 * pos)) & 0xF
        np.testing.assert_array_equal(nib.astype(np.int32) - 8, q.reshape(N, K // 8, 8)[..., k_in])
    # every value of the int4 range round-trips (reference limits: test/data_type/test_int4.cpp:19-52)
    allv = np.tile(np.arange(-8, 8, dtype=np.int8), 4).reshape(1, 64)
    np.testing.assert_array_equal(port.unpack_int4(port.pack_int4(allv, 64, 1), 64, 1), allv)
    # rejected: K not a multiple of 8, value out of range
    import pytest
    with pytest.raises(ValueError):
        port.pack_int4(q[:, :60].copy(), 60, N)
    bad = q.copy()
    bad[0, 0] = 8
    with pytest.raises(ValueError):
        port.pack_int4(bad, K, N)


def test_splitk_partition_and_reduce(port):
    # KRead = ceil(K / (KBatch*v)) * v ; the last slice takes the remainder
    assert port.splitk_kread(65536, 8, 64) == 8192
    assert port.splitk_kread(437, 3, 8) == 152
    assert port.splitk_kread(4096, 5, 64) == 832
    rng = np.random.default_rng(3)
    M, N, K, kb = 16, 24, 256, 3
    a = np.round(rng.uniform(-2, 2, (M, K))).astype(np.float16).view(np.uint16)
    b = rng.integers(-8, 8, (N, K)).astype(np.int8)
    s = rng.integers(1, 3, N).astype(np.float16).view(np.uint16)
    kr = port.splitk_kread(K, kb, 64)
    parts = np.stack([port.fpAintB_partial(a, b, s, M, N, K, K, z * kr, min(K, (z + 1) * kr)) for z in range(kb)])
    full = port.fpAintB_gemm_f16(a, b, s, M, N, K, K, K, N)
    # integer-valued inputs: partial sums are exact, so split-K equals the one-pass result bit for bit
    np.testing.assert_array_equal(port.splitk_reduce_f16(parts), full)


def test_splitk_unscaled_bf16_i8_cases(port, golden_gemm):
    """example/35's problem (bf16 A, row-major int8 B, no scale): the port's loop against the reference's
    ReferenceGemm<bhalf, int8, bhalf, float> outputs, bit for bit."""
    g = golden_gemm
    M, N, K = 64, 128, 512
    got = port.gemm_bf16_i8(g["sk_a_int"], g["sk_brow"], None, M, N, K, K, N, 1, N, 0)
    np.testing.assert_array_equal(got, g["sk_c_int"])
    got = port.gemm_bf16_i8(g["sk_a_dec"], g["sk_brow_full"], None, M, N, K, K, N, 1, N, 0)
    np.testing.assert_array_equal(got, g["sk_c_dec"])


def test_bf16_i8_full_matrix_d_cases(port, golden_gemm):
    """D tensors handed over as full (M, N) matrices (row stride = leading dimension) against the reference's
    functors applied to d_m_n(m, n)."""
    g = golden_gemm
    M, N, K = 48, 64, 128
    for ep in (3, 4, 5, 6):
        got = port.gemm_bf16_i8_dmat(g["bf16_a"], g["bf16_bcol"], g["bf16_s"], M, N, K, K, 1, K, N, ep, g["bf16_dm0"], N)
        np.testing.assert_array_equal(got, g[f"bf16_dmat_e_col_ep{ep}"])
    got = port.gemm_bf16_i8_dmat(g["bf16_a"], g["bf16_bcol"], None, M, N, K, K, 1, K, N, 2, g["bf16_dm0"], N,
                                 g["bf16_dm1"], N + 8)
    np.testing.assert_array_equal(got, g["bf16_dmat_e_col_ep2"])
    # ldd = 0 is the per-column form
    a = port.gemm_bf16_i8_dmat(g["bf16_a"], g["bf16_bcol"], g["bf16_s"], M, N, K, K, 1, K, N, 3, g["bf16_d0"], 0)
    b = port.gemm_bf16_i8(g["bf16_a"], g["bf16_bcol"], g["bf16_s"], M, N, K, K, 1, K, N, 3, g["bf16_d0"], g["bf16_d1"])
    np.testing.assert_array_equal(a, b)


GROUPED_CASES = ("grp64_i4", "grp128_i4", "grp64_i8", "grp_ragged_i4", "grp128_int_i4")


def test_grouped_scale_cases(port, golden_gemm):
    """Group-wise weight scales: the checker against the reference's ReferencefpAintBGemm run on the (K, N) scale
    tensor whose rows repeat per group (groups of 64 / 128 / 32 with a partial last group, padded scale rows)."""
    g = golden_gemm
    for tag in GROUPED_CASES:
        M, N, K, lda, ldb, ldc, group_k, ld_scale = (int(v) for v in g[f"{tag}_shape"])
        c = port.fpAintB_gemm_f16_grouped(g[f"{tag}_a"], g[f"{tag}_b"], g[f"{tag}_s"], M, N, K, lda, ldb, ldc, group_k,
                                          ld_scale)
        np.testing.assert_array_equal(c, g[f"{tag}_c"])
    # one group spanning all of K is the per-channel GEMM
    M, N, K, lda, ldb, ldc = (int(v) for v in g["fpaintb_real_shape"])
    c = port.fpAintB_gemm_f16_grouped(g["fpaintb_real_a"], g["fpaintb_real_b"], g["fpaintb_real_s"][None, :], M, N, K,
                                      lda, ldb, ldc, K, N)
    np.testing.assert_array_equal(c, g["fpaintb_real_c"])


def test_multiply_add_bit_pattern(port, golden_gemm):
    """MultiplyAdd<bhalf_t, float, bhalf_t, bhalf_t>: the reference functor on random operands and inside the bf16 x int8
    GEMM ("multiply" instance form, per-column and full-matrix D)."""
    g = golden_gemm
    for c, a, b, e in zip(g["mad_c"], g["mad_d0"], g["mad_d1"], g["mad_e"]):
        assert port.multiply_add_bf16(float(c), int(a), int(b)) == int(e)
    M, N, K = 48, 64, 128
    brow = np.ascontiguousarray(g["bf16_bcol"].T)
    e = port.gemm_bf16_i8(g["bf16_a"], brow, None, M, N, K, K, N, 1, N, 7, g["bf16_s"], g["bf16_d1"])
    np.testing.assert_array_equal(e, g["bf16_mul_e_row_ep7"])
    e = port.gemm_bf16_i8_dmat(g["bf16_a"], brow, None, M, N, K, K, N, 1, N, 7, g["bf16_dm0"], N, g["bf16_dm1"], N + 8)
    np.testing.assert_array_equal(e, g["bf16_dmat_e_row_ep7"])


def test_convert_bf16_rtn(port, golden_scalars, golden_gemm):
    """ConvertBF16RTN (unary_element_wise_operation.hpp:236-251 -> type_convert.hpp:507-549): the reference functor on
    random values, exact ties, carries into Inf, denormals, Inf / NaN patterns; and as the C operation of the bf16 x int8
    GEMM (Multiply on B, unscaled row-major B, and a half-integer set whose sums hit ties)."""
    s, g = golden_scalars, golden_gemm
    for bits, want in zip(s["rtn_in_bits"], s["rtn_bf16"]):
        x = float(np.uint32(bits).view(np.float32))
        if x != x:
            # a NaN's payload does not survive the float -> double -> float round trip of the ctypes call on every
            # host; the fixture keeps the reference's answer for the record, the port is checked on its class
            assert port.bf16_convert_rtn(x) & 0x7F80 == 0x7F80 and port.bf16_convert_rtn(x) & 0x7F
            continue
        assert port.bf16_convert_rtn(x) == int(want), hex(int(bits))
    M, N, K = 48, 64, 128
    brow = np.ascontiguousarray(g["bf16_bcol"].T)
    e = port.gemm_bf16_i8(g["bf16_a"], g["bf16_bcol"], g["bf16_s"], M, N, K, K, 1, K, N, 8)
    np.testing.assert_array_equal(e, g["bf16_e_col_ep8"])
    assert (e != g["bf16_e_col_ep0"]).any()  # differs from truncation somewhere
    e = port.gemm_bf16_i8(g["bf16_a"], brow, None, M, N, K, K, N, 1, N, 8)
    np.testing.assert_array_equal(e, g["bf16_mul_e_row_ep8"])
    e = port.gemm_bf16_i8(g["rtn_a_tie"], brow, None, M, N, K, K, N, 1, N, 8)
    np.testing.assert_array_equal(e, g["bf16_tie_e_row_ep8"])


BF16_GROUPED_CASES = ("bgrp64_i8", "bgrp128_i4", "bgrp32_i8_add", "bgrp64_i4_rtn", "bgrp128_int_i4")


def test_grouped_scale_cases_bf16(port, golden_gemm):
    """Group-wise weight scales with bf16 activations: the checker against the reference's Multiply-on-B functor applied
    with a B1 tensor whose rows repeat per group, ReferenceGemm and the epilogue functors (PassThrough, FastGelu, Add,
    ConvertBF16RTN)."""
    g = golden_gemm
    for tag in BF16_GROUPED_CASES:
        M, N, K, group_k, ld_b1, ep = (int(v) for v in g[f"{tag}_shape"])
        e = port.gemm_bf16_i8_grouped(g[f"{tag}_a"], g[f"{tag}_b"], g[f"{tag}_s"], M, N, K, K, 1, K, N, ep, group_k, ld_b1,
                                      g[f"{tag}_d0"])
        np.testing.assert_array_equal(e, g[f"{tag}_e"])
    # one group spanning all of K is the per-channel GEMM
    M, N, K = 48, 64, 128
    e = port.gemm_bf16_i8_grouped(g["bf16_a"], g["bf16_bcol"], g["bf16_s"][None, :], M, N, K, K, 1, K, N, 0, K, N)
    np.testing.assert_array_equal(e, g["bf16_e_col_ep0"])


@pytest.mark.parametrize("tag", ["k320", "k437"])
def test_splitk_reference_shapes_golden(port, tag):
    """K = 320 and K = 437 of the reference's split-K test (test/gemm_split_k/test_gemm_splitk_ut_cases.inc): the plain-C
    checker reproduces the reference's ReferenceGemm<bhalf, int8, bhalf, float> outputs bit for bit."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "splitk_shapes.npz"))
    M, N, K = (int(v) for v in g[f"{tag}_shape"])
    for kind, b in (("int", g[f"{tag}_brow"]), ("dec", g[f"{tag}_brow_full"])):
        q = np.ascontiguousarray(b.T)      # (N, K): K contiguous per column
        got = port.gemm_bf16_i8(g[f"{tag}_a_{kind}"], q, None, M, N, K, K, 1, K, N, 0)
        np.testing.assert_array_equal(got, g[f"{tag}_c_{kind}"])


ABSCALE_TAGS = ["ab128_i4_bf16", "ab128_i4_f16_pow2", "ab_small_blocks_i8", "ab_ragged_i4"]


@pytest.mark.parametrize("tag", ABSCALE_TAGS)
def test_f8_block_scaled_gemm_golden(port, tag):
    """fp8 activations x int4 / int8 weights with block-wise scales of both operands: the plain-C checker against outputs of
    the reference's own host check (example/65 ..._fp8_ab_scale.cpp:262-283), bit for bit."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "abscale_cases.npz"))
    M, N, K, bm, bn, bk, ld_sa, ld_sb, bf = (int(v) for v in g[f"{tag}_shape"])
    got = port.gemm_f8_i8_abscale(g[f"{tag}_a"], g[f"{tag}_q"], g[f"{tag}_sa"], g[f"{tag}_sb"], M, N, K, K, K, N, (bm, bn, bk),
                                  ld_sa, ld_sb, e_bf16=bool(bf))
    np.testing.assert_array_equal(got, g[f"{tag}_e"])
