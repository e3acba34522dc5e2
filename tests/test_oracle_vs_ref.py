"""Live cross-check of the plain-C checker against the reference's own host templates
(oracle/_ref/libck_ref.so).  Skipped where neither the reference tree nor a prebuilt
oracle/_ref is present."""
import numpy as np


def test_scalar_functions_random(port, ref):
    rng = np.random.default_rng(7)
    fl = (rng.uniform(-1, 1, 4000) * rng.choice([1e-8, 1e-6, 6e-5, 1e-3, 1, 100, 7e4], 4000)).astype(np.float32)
    for f in fl:
        f = float(f)
        assert port.f32_to_f16(f) == ref.f32_to_f16(f)
        assert port.f32_to_bf16(f) == ref.f32_to_bf16(f)
        assert np.float32(port.fastgelu(f)).view(np.uint32) == np.float32(ref.fastgelu(f)).view(np.uint32)
    for q in range(-128, 128):
        for s in rng.integers(0, 0x7C00, 24):
            assert port.dequant_f16(q, int(s)) == ref.dequant_f16(q, int(s))
            assert port.multiply_bf16(q, int(s)) == ref.multiply_bf16(q, int(s))
    # bf16_convert_rtn on arbitrary bit patterns (NaN and Inf patterns included)
    for u in rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32):
        f = float(np.uint32(u).view(np.float32))
        if f == f:  # a Python float cannot carry a NaN payload through ctypes unchanged
            assert port.bf16_convert_rtn(f) == ref.bf16_convert_rtn(f)
    for x in rng.integers(-200000, 200000, 3000):
        assert port.act_mul_clamp(0.03, int(x)) == ref.act_mul_clamp(0.03, int(x))
        assert port.act_mul_clamp_from_float(0.03, float(x)) == ref.act_mul_clamp_from_float(0.03, float(x))


def test_fills_long(port, ref):
    for dt in ("f16", "bf16", "i8", "f32"):
        for iv in (False, True):
            a = port.fill_uniform(20000, dt, -1.0, 1.0, iv)
            b = ref.fill_uniform(20000, dt, -1.0, 1.0, iv)
            np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8))


def test_fpaintb_random_shapes(port, ref):
    rng = np.random.default_rng(5)
    for M, N, K in ((7, 9, 40), (64, 64, 512), (1, 130, 64)):
        a = rng.uniform(-1, 1, (M, K)).astype(np.float16).view(np.uint16)
        b = rng.integers(-8, 8, (N, K)).astype(np.int8)
        s = rng.uniform(2.0 ** -8, 2.0 ** -4, N).astype(np.float16).view(np.uint16)
        np.testing.assert_array_equal(port.fpAintB_gemm_f16(a, b, s, M, N, K, K, K, N),
                                      ref.fpAintB_gemm_f16(a, b, s, M, N, K, K, K, N))


def test_bf16_and_f8_random(port, ref):
    rng = np.random.default_rng(9)
    M, N, K = 24, 40, 96
    a = (rng.uniform(-1, 1, (M, K)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    s = (rng.uniform(0.004, 0.06, N).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    b = rng.integers(-128, 128, (N, K)).astype(np.int8)
    d0 = (rng.uniform(0.5, 1.5, N).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    d1 = (rng.uniform(-0.5, 0.5, N).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    for ep in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        np.testing.assert_array_equal(port.gemm_bf16_i8(a, b, s, M, N, K, K, 1, K, N, ep, d0, d1),
                                      ref.gemm_bf16_i8(a, b, s, M, N, K, K, 1, K, N, ep, d0, d1))
        np.testing.assert_array_equal(port.gemm_bf16_i8(a, b, None, M, N, K, K, 1, K, N, ep, d0, d1),
                                      ref.gemm_bf16_i8(a, b, None, M, N, K, K, 1, K, N, ep, d0, d1))
    a8 = rng.integers(0, 256, (M, K)).astype(np.uint8)
    a8[(a8 == 0x80)] = 0
    b4 = rng.integers(-8, 8, (N, K)).astype(np.int8)
    f0 = rng.uniform(0, 1, N).astype(np.float32)
    f1 = rng.uniform(0, 1, M).astype(np.float32)
    for gelu in (False, True):
        np.testing.assert_array_equal(port.gemm_f8_i8_mm(a8, b4, f0, f1, M, N, K, K, K, N, gelu),
                                      ref.gemm_f8_i8_mm(a8, b4, f0, f1, M, N, K, K, K, N, gelu))


def test_i8_bias_requant_random(port, ref):
    rng = np.random.default_rng(21)
    for M, N, K, lo, hi, rq in ((9, 17, 50, -128, 128, 0.002), (32, 8, 300, -20, 21, 0.01), (5, 64, 1024, -128, 128, 0.03)):
        a = rng.integers(lo, hi, (M, K)).astype(np.int8)
        b = rng.integers(lo, hi, (N, K)).astype(np.int8)
        bias = rng.integers(-5000, 5001, N).astype(np.int32)
        for relu in (0, 1):
            for bs in (None, bias):
                np.testing.assert_array_equal(port.gemm_i8_bias_requant(a, b, bs, M, N, K, K, K, N, rq, relu),
                                              ref.gemm_i8_bias_requant(a, b, bs, M, N, K, K, K, N, rq, relu))


def test_grouped_scales_random(port, ref):
    rng = np.random.default_rng(21)
    for M, N, K, g in ((5, 12, 96, 32), (16, 40, 256, 64), (3, 8, 200, 128), (9, 16, 64, 1)):
        groups = (K + g - 1) // g
        a = rng.uniform(-1, 1, (M, K)).astype(np.float16).view(np.uint16)
        b = rng.integers(-128, 128, (N, K)).astype(np.int8)
        s = rng.uniform(2.0 ** -8, 2.0 ** -4, (groups, N + 4)).astype(np.float16).view(np.uint16)
        np.testing.assert_array_equal(port.fpAintB_gemm_f16_grouped(a, b, s, M, N, K, K, K, N, g, N + 4),
                                      ref.fpAintB_gemm_f16_grouped(a, b, s, M, N, K, K, K, N, g, N + 4))


def test_grouped_scales_bf16_random(port, ref):
    rng = np.random.default_rng(13)
    for M, N, K, gk, ep in ((9, 24, 160, 32, 0), (20, 40, 256, 128, 1), (5, 16, 96, 64, 3), (12, 32, 128, 64, 8)):
        a = (rng.uniform(-1, 1, (M, K)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        b = rng.integers(-128, 128, (N, K)).astype(np.int8)
        s = (rng.uniform(0.004, 0.06, ((K + gk - 1) // gk, N)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        d0 = (rng.uniform(-0.5, 0.5, N).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        np.testing.assert_array_equal(port.gemm_bf16_i8_grouped(a, b, s, M, N, K, K, 1, K, N, ep, gk, N, d0),
                                      ref.gemm_bf16_i8_grouped(a, b, s, M, N, K, K, 1, K, N, ep, gk, N, d0))


def test_f8_block_scaled_random(port, ref):
    """Block-wise scales of both operands (example/65's fp8_ab_scale host check): port == reference on random shapes,
    block sizes and paddings."""
    rng = np.random.default_rng(65)
    for _ in range(12):
        M, N, K = (int(v) for v in rng.integers(1, 70, 3))
        K = max(K, 2)
        block = tuple(int(v) for v in rng.choice([1, 4, 16, 32, 128], 3))
        kb = (K + block[2] - 1) // block[2]
        ld_sa, ld_sb = kb + int(rng.integers(0, 3)), kb + int(rng.integers(0, 3))
        a = rng.integers(0, 256, (M, K)).astype(np.uint8)
        a[a == 0x80] = 0
        q = rng.integers(-128, 128, (N, K)).astype(np.int8)
        sa = rng.uniform(-1, 1, ((M + block[0] - 1) // block[0], ld_sa)).astype(np.float32)
        sb = rng.uniform(-1, 1, ((N + block[1] - 1) // block[1], ld_sb)).astype(np.float32)
        for bf in (False, True):
            x = port.gemm_f8_i8_abscale(a, q, sa, sb, M, N, K, K, K, N, block, ld_sa, ld_sb, e_bf16=bf)
            y = ref.gemm_f8_i8_abscale(a, q, sa, sb, M, N, K, K, K, N, block, ld_sa, ld_sb, e_bf16=bf)
            np.testing.assert_array_equal(x, y)
