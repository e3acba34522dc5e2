#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own host code.

Run in the container that mounts the reference tree (it builds oracle/_ref/libck_ref.so from
the reference headers and calls it through ctypes).  The outputs are plain ``.npz`` files holding
inputs and the reference's outputs only -- data, no code -- and are committed, so the parity
tests also run where the reference is absent (the GPU box).

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def f16_bits(x: np.ndarray) -> np.ndarray:
    return np.asarray(x, dtype=np.float16).view(np.uint16)


def bf16_bits_trunc(x: np.ndarray) -> np.ndarray:
    return (np.asarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def scalars(ref) -> dict:
    rng = np.random.default_rng(20250301)
    out = {}
    # fp16 dequant: every int8 value x a spread of scales (incl. subnormal / large / negative)
    scales = np.concatenate([
        f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, 24)),
        f16_bits(rng.uniform(-1, 1, 16)),
        np.array([0x3555, 0x2E66, 0xB99A, 0x0400, 0x0001, 0x03FF, 0x7BFF, 0x3C00, 0xBC00, 0x0000], np.uint16),
    ]).astype(np.uint16)
    q = np.arange(-128, 128, dtype=np.int16)
    out["dequant_f16_scales"] = scales
    out["dequant_f16"] = np.array([[ref.dequant_f16(int(v), int(s)) for v in q] for s in scales], np.uint16)
    bscales = np.concatenate([
        bf16_bits_trunc(rng.uniform(2.0 ** -8, 2.0 ** -4, 24)),
        bf16_bits_trunc(rng.uniform(-2, 2, 16)),
        np.array([0x3EAB, 0x3F9A, 0x3C23, 0x3F80, 0x0080, 0x7F7F], np.uint16),
    ]).astype(np.uint16)
    out["multiply_bf16_scales"] = bscales
    out["multiply_bf16"] = np.array([[ref.multiply_bf16(int(v), int(s)) for v in q] for s in bscales], np.uint16)
    # conversions
    fx = np.concatenate([
        rng.uniform(-1, 1, 512) * rng.choice([1e-8, 1e-6, 6e-5, 1e-3, 1.0, 100.0, 7e4], 512),
        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001,
                  2.0 ** -14, 1.00048828125, 1.0009765625 + 2.0 ** -11]),
    ]).astype(np.float32)
    out["f32_in"] = fx
    out["f32_to_f16"] = np.array([ref.f32_to_f16(float(v)) for v in fx], np.uint16)
    out["f32_to_bf16"] = np.array([ref.f32_to_bf16(float(v)) for v in fx], np.uint16)
    out["f16_to_f32"] = np.array([ref.f16_to_f32(h) for h in range(0, 65536, 7)], np.float32)
    out["f8_to_f32"] = np.array([ref.f8_to_f32(v) for v in range(256)], np.float32)
    # FastGelu host <float,float>
    gx = np.concatenate([rng.normal(size=1024) * 3.0, np.array([-3, -1, 0.5, 1, 3, 10, 0, -20, 20])]).astype(np.float32)
    out["fastgelu_in"] = gx
    out["fastgelu"] = np.array([ref.fastgelu(float(v)) for v in gx], np.float32)
    # Activation_Mul_Clamp(0.03) over accumulator values (int32 form and float-accumulator form)
    acc = np.concatenate([np.arange(-4400, 4400, 7), np.array([33, 34, -34, 50, 84, 4233, 4234, -4266, -4267,
                                                                100000, -100000, 2 ** 24, -(2 ** 24)])]).astype(np.int32)
    out["amc_in"] = acc
    out["amc_0p03"] = np.array([ref.act_mul_clamp(0.03, int(v)) for v in acc], np.int8)
    out["amc_from_float_0p03"] = np.array([ref.act_mul_clamp_from_float(0.03, float(v)) for v in acc], np.int8)
    # MultiplyMultiply / MultiplyAddFastGelu
    c = rng.normal(size=256).astype(np.float32) * 50
    d0 = rng.uniform(0, 1, 256).astype(np.float32)
    d1 = rng.uniform(0, 1, 256).astype(np.float32)
    out["mm_c"], out["mm_d0"], out["mm_d1"] = c, d0, d1
    out["mm_f16"] = np.array([ref.multiply_multiply_f16(float(a), float(b), float(e)) for a, b, e in zip(c, d0, d1)], np.uint16)
    out["mm_bf16"] = np.array([ref.multiply_multiply_bf16(float(a), float(b), float(e)) for a, b, e in zip(c, d0, d1)], np.uint16)
    b0 = bf16_bits_trunc(d0)
    b1 = bf16_bits_trunc(d1 - 0.5)
    out["mafg_d0"], out["mafg_d1"] = b0, b1
    out["mafg_bf16"] = np.array([ref.multiply_add_fastgelu_bf16(float(a), int(x), int(y)) for a, x, y in zip(c, b0, b1)], np.uint16)
    # bf16_convert_rtn (type_convert.hpp:507-549) through the functor ConvertBF16RTN: random values, exact ties on both
    # parities, carries into the exponent and into Inf, denormals, Inf and NaN patterns (payload in either half)
    rb = np.concatenate([
        (rng.uniform(-1, 1, 512) * rng.choice([1e-30, 1e-6, 1.0, 300.0, 3e38], 512)).astype(np.float32).view(np.uint32),
        rng.integers(0, 2 ** 32, 512, dtype=np.uint64).astype(np.uint32),
        np.array([0x3F808000, 0x3F818000, 0x3F808001, 0x3F807FFF, 0x3FFF8000, 0x3FFFFFFF, 0x7F7F8000, 0x7F7FFFFF,
                  0x7F7F7FFF, 0x7F800000, 0xFF800000, 0x7FC00000, 0x7F800001, 0xFF80FFFF, 0x7FFFFFFF, 0x7F810000,
                  0x00008000, 0x00007FFF, 0x00018000, 0x007F8000, 0x007FFFFF, 0x80000000, 0x00000000, 0x80008000,
                  0xC02DF854, 0x477FE000], np.uint32),
    ]).astype(np.uint32)
    out["rtn_in_bits"] = rb
    out["rtn_bf16"] = np.array([ref.bf16_convert_rtn(float(v)) if np.isfinite(v) or True else 0 for v in rb.view(np.float32)], np.uint16)
    # generators: first values of every (dtype, mode) the drivers use
    for dt in ("f16", "bf16", "i8", "f32"):
        for iv in (0, 1):
            out[f"fill_{dt}_{iv}"] = ref.fill_uniform(96, dt, -1.0, 1.0, bool(iv))
    out["fill_i8_int4range"] = ref.fill_uniform(96, "i8", -8.0, 7.0, True)
    out["gen2_i8_seed1"] = ref.generator2_i8(8, 12, 12, 1, -128, 127, seed=1)
    out["gen2_i8_seed1_col"] = ref.generator2_i8(8, 12, 1, 8, -5, 5, seed=1)
    return out


def gemm_cases(ref) -> dict:
    rng = np.random.default_rng(11939)
    out = {}

    # ---- fp16 A x int4-range B (config C2 semantics, small) ---------------------------------
    def fpaintb(tag, M, N, K, lda, ldb, ldc, integer, qlo, qhi):
        if integer:
            a = f16_bits(np.round(rng.uniform(-1, 1, (M, lda))))
            s = f16_bits(rng.integers(1, 4, N).astype(np.float32))
        else:
            a = f16_bits(rng.uniform(-1, 1, (M, lda)))
            s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
        b = rng.integers(qlo, qhi + 1, (N, ldb)).astype(np.int8)  # column-major (K,N): row n holds K values
        c = ref.fpAintB_gemm_f16(a, b, s, M, N, K, lda, ldb, ldc)
        out[f"{tag}_shape"] = np.array([M, N, K, lda, ldb, ldc], np.int32)
        out[f"{tag}_a"], out[f"{tag}_b"], out[f"{tag}_s"], out[f"{tag}_c"] = a, b, s, c

    fpaintb("fpaintb_real", 64, 48, 128, 128, 128, 48, False, -8, 7)
    fpaintb("fpaintb_int", 128, 128, 256, 256, 256, 128, True, -8, 7)
    fpaintb("fpaintb_ragged", 33, 17, 72, 80, 88, 24, False, -8, 7)
    fpaintb("fpaintb_i8", 48, 32, 64, 64, 64, 32, False, -128, 127)

    # example/64 exactly (init_method 1 and 2, tiny): A, quant B and scale each from a fresh
    # mt19937(11939) -- run_gemm_example.inc:36-49
    for init in (1, 2):
        M, N, K = 32, 32, 64
        a = ref.fill_uniform(M * K, "f16", -1.0, 1.0, init == 1).reshape(M, K)
        b = ref.fill_uniform(K * N, "i8", -1.0, 1.0, init == 1).reshape(N, K)
        s = ref.fill_uniform(N, "f16", -1.0, 1.0, init == 1)
        out[f"ex64_init{init}_c"] = ref.fpAintB_gemm_f16(a, b, s, M, N, K, K, K, N)

    # ---- C1: example/14 host verify path, M=N=K=256, srand(1), GeneratorTensor_2{-128,127} ---
    M = N = K = 256
    a = ref.generator2_i8(M, K, K, 1, -128, 127, seed=1)      # row-major A, drawn first
    b = ref.generator2_i8(K, N, 1, K, -128, 127, seed=-1)     # column-major B continues the stream
    e = ref.gemm_i8_requant(a, b, M, N, K, K, K, N, 0.03)
    out["c1_a_head"], out["c1_b_head"] = a[:64].copy(), b[:64].copy()
    out["c1_e"] = e.reshape(M, N)

    # a second int8 case with small values and ragged strides
    M, N, K = 40, 24, 96
    a = rng.integers(-5, 6, (M, 104)).astype(np.int8)
    b = rng.integers(-5, 6, (N, 100)).astype(np.int8)
    out["i8_small_a"], out["i8_small_b"] = a, b
    out["i8_small_e"] = ref.gemm_i8_requant(a, b, M, N, K, 104, 100, 32, 0.25)

    # ---- bf16 A x int8 B (config C3 semantics, small), three epilogues, both B layouts -------
    M, N, K = 48, 64, 128
    a = bf16_bits_trunc(rng.uniform(-1, 1, (M, K)))
    s = bf16_bits_trunc(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    bcol = rng.integers(-128, 128, (N, K)).astype(np.int8)    # (k,n) at n*K + k
    brow = np.ascontiguousarray(bcol.T)                        # (k,n) at k*N + n
    d0 = bf16_bits_trunc(rng.uniform(0.5, 1.5, N))
    d1 = bf16_bits_trunc(rng.uniform(-0.5, 0.5, N))
    out["bf16_a"], out["bf16_s"], out["bf16_bcol"], out["bf16_d0"], out["bf16_d1"] = a, s, bcol, d0, d1
    for ep in (0, 1, 2):
        out[f"bf16_e_col_ep{ep}"] = ref.gemm_bf16_i8(a, bcol, s, M, N, K, K, 1, K, N, ep, d0, d1)
    out["bf16_e_row_ep1"] = ref.gemm_bf16_i8(a, brow, s, M, N, K, K, N, 1, N, 1, d0, d1)
    # the binary epilogues of the multi-ABD instance family (3 Add, 4 AddFastGelu, 5 Multiply, 6 MultiplyFastGelu),
    # with Multiply on B (b1 = s) and for the "multiply" instances (b1 absent, row-major B as in client_example/30)
    for ep in (3, 4, 5, 6):
        out[f"bf16_e_col_ep{ep}"] = ref.gemm_bf16_i8(a, bcol, s, M, N, K, K, 1, K, N, ep, d0, d1)
    for ep in (0, 2, 5, 6):
        out[f"bf16_mul_e_row_ep{ep}"] = ref.gemm_bf16_i8(a, brow, None, M, N, K, K, N, 1, N, ep, s, d1)

    # ---- fp8 (FNUZ bytes) A x int4-range B + MultiplyMultiply (+FastGelu) (config C5, small) --
    M, N, K = 48, 64, 128
    a8 = rng.integers(0, 256, (M, K)).astype(np.uint8)
    a8[(a8 == 0x80) | (a8 == 0x7F) | (a8 == 0xFF)] = 0x38     # input contract: no NaN / +-max codes
    a8 = np.where((a8 & 0x7F) > 0x57, a8 & 0xBF, a8).astype(np.uint8)  # keep |a| modest
    b4 = rng.integers(-8, 8, (N, K)).astype(np.int8)
    f0 = rng.uniform(0, 1, N).astype(np.float32)
    f1 = rng.uniform(0, 1, M).astype(np.float32)
    out["f8_a"], out["f8_b"], out["f8_d0"], out["f8_d1"] = a8, b4, f0, f1
    out["f8_e_mm"] = ref.gemm_f8_i8_mm(a8, b4, f0, f1, M, N, K, K, K, N, False)
    out["f8_e_mm_gelu"] = ref.gemm_f8_i8_mm(a8, b4, f0, f1, M, N, K, K, K, N, True)

    # ---- int8 requantisation with Relu and/or an int32 bias (example/14 bias + relu verify path) -------------
    # inputs drawn like the example's: srand(1), then A, B and the bias from GeneratorTensor_2{-128,127} in that order
    M = N = K = 128
    a = ref.generator2_i8(M, K, K, 1, -128, 127, seed=1)
    b = ref.generator2_i8(K, N, 1, K, -128, 127, seed=-1)
    bias = ref.generator2_i8(1, N, N, 1, -128, 127, seed=-1).astype(np.int32)
    out["i8b_bias"] = bias
    for relu in (0, 1):
        out[f"i8b_e_relu{relu}"] = ref.gemm_i8_bias_requant(a, b, None, M, N, K, K, K, N, 0.03, relu)
        out[f"i8b_e_bias_relu{relu}"] = ref.gemm_i8_bias_requant(a, b, bias, M, N, K, K, K, N, 0.03, relu)
    # the small ragged case above with a bias that matters at its accumulator range
    M, N, K = 40, 24, 96
    bias_s = rng.integers(-200, 201, N).astype(np.int32)
    out["i8_small_bias"] = bias_s
    for relu in (0, 1):
        out[f"i8_small_e_bias_relu{relu}"] = ref.gemm_i8_bias_requant(
            out["i8_small_a"], out["i8_small_b"], bias_s, M, N, K, 104, 100, 32, 0.25, relu)
        out[f"i8_small_e_relu{relu}"] = ref.gemm_i8_bias_requant(
            out["i8_small_a"], out["i8_small_b"], None, M, N, K, 104, 100, 32, 0.25, relu)

    # ---- example/35's split-K problem: bf16 A x int8 B (row-major (K, N), no scale) -> bf16, the reference's
    # ReferenceGemm<bhalf, int8, bhalf, float> over the whole K.  Integer set as its init_method 2 draws
    # (GeneratorTensor_2{-2, 2}: values -2 .. 1), and a decimal set.
    M, N, K = 64, 128, 512
    brow = rng.integers(-2, 2, (K, N)).astype(np.int8)
    a_int = bf16_bits_trunc(rng.integers(-2, 2, (M, K)).astype(np.float32))
    a_dec = bf16_bits_trunc(rng.uniform(-0.5, 0.5, (M, K)))
    brow_full = rng.integers(-128, 128, (K, N)).astype(np.int8)
    out["sk_a_int"], out["sk_a_dec"], out["sk_brow"], out["sk_brow_full"] = a_int, a_dec, brow, brow_full
    out["sk_c_int"] = ref.gemm_bf16_i8_unscaled(a_int, brow, M, N, K, K, N, 1, N)
    out["sk_c_dec"] = ref.gemm_bf16_i8_unscaled(a_dec, brow_full, M, N, K, K, N, 1, N)

    # ---- the bf16 x int8 case above with D handed over as a full (M, N) matrix (client_example/30's bias form) ----
    M, N, K = 48, 64, 128
    dm0 = bf16_bits_trunc(rng.uniform(0.5, 1.5, (M, N)))
    dm1 = bf16_bits_trunc(rng.uniform(-0.5, 0.5, (M, N + 8)))      # padded rows: ldd1 = N + 8
    out["bf16_dm0"], out["bf16_dm1"] = dm0, dm1
    for ep in (3, 4, 5, 6):
        out[f"bf16_dmat_e_col_ep{ep}"] = ref.gemm_bf16_i8_dmat(out["bf16_a"], out["bf16_bcol"], out["bf16_s"], M, N, K, K, 1, K,
                                                               N, ep, dm0, N)
    out["bf16_dmat_e_col_ep2"] = ref.gemm_bf16_i8_dmat(out["bf16_a"], out["bf16_bcol"], None, M, N, K, K, 1, K, N, 2, dm0, N,
                                                       dm1, N + 8)

    # ---- group-wise weight scales (one row of N fp16 scales per group_k consecutive k): the reference's
    # ReferencefpAintBGemm given the (K, N) scale tensor whose rows repeat per group.  int4-range and int8 weights,
    # groups of 64 and 128, a padded scale row (ld_scale = N + 8) and a ragged case with a partial last group.
    def grouped(tag, M, N, K, lda, ldb, ldc, group_k, ld_scale, qlo, qhi, integer=False):
        groups = (K + group_k - 1) // group_k
        if integer:
            a = f16_bits(np.round(rng.uniform(-2, 2, (M, lda))))
            s = f16_bits(rng.integers(1, 4, (groups, ld_scale)).astype(np.float32))
        else:
            a = f16_bits(rng.uniform(-1, 1, (M, lda)))
            s = f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, (groups, ld_scale)))
        b = rng.integers(qlo, qhi + 1, (N, ldb)).astype(np.int8)
        out[f"{tag}_shape"] = np.array([M, N, K, lda, ldb, ldc, group_k, ld_scale], np.int32)
        out[f"{tag}_a"], out[f"{tag}_b"], out[f"{tag}_s"] = a, b, s
        out[f"{tag}_c"] = ref.fpAintB_gemm_f16_grouped(a, b, s, M, N, K, lda, ldb, ldc, group_k, ld_scale)

    grouped("grp64_i4", 48, 64, 256, 256, 256, 64, 64, 64, -8, 7)
    grouped("grp128_i4", 48, 64, 256, 256, 256, 64, 128, 72, -8, 7)
    grouped("grp64_i8", 32, 48, 128, 128, 128, 48, 64, 48, -128, 127)
    grouped("grp_ragged_i4", 33, 20, 80, 88, 96, 24, 32, 20, -8, 7)
    grouped("grp128_int_i4", 128, 128, 512, 512, 512, 128, 128, 128, -8, 7, integer=True)

    # ---- MultiplyAdd<bhalf, float, bhalf, bhalf> (element_wise_operation.hpp:200-231: integer arithmetic on the 16-bit
    # patterns) on the bf16 x int8 case above: the "multiply" instance form (row-major B, no B scale), per-column and
    # full-matrix D operands; and a table of the functor itself
    M, N, K = 48, 64, 128
    brow = np.ascontiguousarray(out["bf16_bcol"].T)
    out["bf16_mul_e_row_ep7"] = ref.gemm_bf16_i8(out["bf16_a"], brow, None, M, N, K, K, N, 1, N, 7, out["bf16_s"], out["bf16_d1"])
    out["bf16_dmat_e_row_ep7"] = ref.gemm_bf16_i8_dmat(out["bf16_a"], brow, None, M, N, K, K, N, 1, N, 7, out["bf16_dm0"], N,
                                                       out["bf16_dm1"], N + 8)
    mc = rng.uniform(-300, 300, 512).astype(np.float32)
    md0 = rng.integers(0, 65536, 512).astype(np.uint16)
    md1 = rng.integers(0, 65536, 512).astype(np.uint16)
    out["mad_c"], out["mad_d0"], out["mad_d1"] = mc, md0, md1
    out["mad_e"] = np.array([ref.multiply_add_bf16(float(c), int(a), int(b)) for c, a, b in zip(mc, md0, md1)], np.uint16)

    # ---- ConvertBF16RTN as the C element-wise operation (example/01_gemm/gemm_xdl_bf16_rtn.cpp:21) on the bf16 x int8
    # case above: Multiply on B (column-major B) and the unscaled row-major form, plus an integer-valued set whose
    # accumulators hit exact ties (sums of odd multiples of 2^-8 scaled into [128, 256) have a set bit 16)
    M, N, K = 48, 64, 128
    out["bf16_e_col_ep8"] = ref.gemm_bf16_i8(out["bf16_a"], out["bf16_bcol"], out["bf16_s"], M, N, K, K, 1, K, N, 8, None, None)
    out["bf16_mul_e_row_ep8"] = ref.gemm_bf16_i8(out["bf16_a"], brow, None, M, N, K, K, N, 1, N, 8, None, None)
    a_tie = bf16_bits_trunc(rng.integers(-3, 4, (M, K)).astype(np.float32) * 0.5)
    out["rtn_a_tie"] = a_tie
    out["bf16_tie_e_row_ep8"] = ref.gemm_bf16_i8(a_tie, brow, None, M, N, K, K, N, 1, N, 8, None, None)

    # ---- group-wise weight scales with bf16 activations: Multiply on B with a B1 tensor whose row k holds the scales of
    # group k / group_k (example/60's (K, N) B1 with repeating rows instead of a zero stride), ReferenceGemm, epilogue.
    # int8 and int4-range weights, groups of 64 / 128 / 32, a padded scale row, PassThrough / FastGelu / Add / RTN.
    def grouped_bf16(tag, M, N, K, group_k, ld_b1, qlo, qhi, ep, integer=False):
        groups = (K + group_k - 1) // group_k
        if integer:
            a = bf16_bits_trunc(np.round(rng.uniform(-2, 2, (M, K))))
            s = bf16_bits_trunc(rng.integers(1, 4, (groups, ld_b1)).astype(np.float32))
        else:
            a = bf16_bits_trunc(rng.uniform(-1, 1, (M, K)))
            s = bf16_bits_trunc(rng.uniform(2.0 ** -8, 2.0 ** -4, (groups, ld_b1)))
        b = rng.integers(qlo, qhi + 1, (N, K)).astype(np.int8)
        d0 = bf16_bits_trunc(rng.uniform(-0.5, 0.5, N))
        out[f"{tag}_shape"] = np.array([M, N, K, group_k, ld_b1, ep], np.int32)
        out[f"{tag}_a"], out[f"{tag}_b"], out[f"{tag}_s"], out[f"{tag}_d0"] = a, b, s, d0
        out[f"{tag}_e"] = ref.gemm_bf16_i8_grouped(a, b, s, M, N, K, K, 1, K, N, ep, group_k, ld_b1, d0, None)

    grouped_bf16("bgrp64_i8", 48, 64, 256, 64, 64, -128, 127, 0)
    grouped_bf16("bgrp128_i4", 48, 64, 256, 128, 72, -8, 7, 1)
    grouped_bf16("bgrp32_i8_add", 33, 24, 96, 32, 24, -128, 127, 3)
    grouped_bf16("bgrp64_i4_rtn", 32, 48, 128, 64, 48, -8, 7, 8)
    grouped_bf16("bgrp128_int_i4", 128, 128, 512, 128, 128, -8, 7, 0, integer=True)
    return out


def splitk_shapes(ref) -> dict:
    """The K extents of the reference's own split-K test that are not whole 64-k tiles (test/gemm_split_k/
    test_gemm_splitk_ut_cases.inc:3-217: K = 320 at N = 512; PaddK K = 437) on example/35's arithmetic:
    ReferenceGemm<bhalf, int8, bhalf, float> over the whole K, B row-major (K, N).  A seed of their own: the file is
    independent of gemm_cases.npz."""
    rng = np.random.default_rng(20250437)
    out = {}
    for tag, (M, N, K) in {"k320": (6, 512, 320), "k437": (127, 512, 437)}.items():
        brow = rng.integers(-2, 2, (K, N)).astype(np.int8)
        brow_full = rng.integers(-128, 128, (K, N)).astype(np.int8)
        a_int = bf16_bits_trunc(rng.integers(-2, 2, (M, K)).astype(np.float32))
        a_dec = bf16_bits_trunc(rng.uniform(-0.5, 0.5, (M, K)))
        out[f"{tag}_shape"] = np.array([M, N, K], np.int32)
        out[f"{tag}_a_int"], out[f"{tag}_a_dec"], out[f"{tag}_brow"], out[f"{tag}_brow_full"] = a_int, a_dec, brow, brow_full
        out[f"{tag}_c_int"] = ref.gemm_bf16_i8_unscaled(a_int, brow, M, N, K, K, N, 1, N)
        out[f"{tag}_c_dec"] = ref.gemm_bf16_i8_unscaled(a_dec, brow_full, M, N, K, K, N, 1, N)
    return out


def abscale_cases(ref) -> dict:
    """fp8 activations x int4 / int8 weights with block-wise scales of both operands: outputs of the reference's own host
    check of DeviceGemmMultiD_ABScale (example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:262-283:
    type_convert<float>(f8_t), fp32 dequantisation of both operands, ReferenceGemm<float, float, float, float>, convert).
    A seed of its own."""
    rng = np.random.default_rng(20250128)
    out = {}

    def case(tag, M, N, K, block, lo, hi, e_bf16, pad=(0, 0), pow2=False):
        bm, bn, bk = block
        kb = (K + bk - 1) // bk
        a = rng.integers(0, 256, (M, K)).astype(np.uint8)
        a[a == 0x80] = 0                      # FNUZ NaN
        if pow2:                              # exact set: activations in {-2 .. 2} (FNUZ bytes), power-of-two scales
            a = np.array([0xC8, 0xC0, 0x00, 0x40, 0x48], np.uint8)[rng.integers(0, 5, (M, K))]
        q = rng.integers(lo, hi + 1, (N, K)).astype(np.int8)
        ld_sa, ld_sb = kb + pad[0], kb + pad[1]
        shape_a, shape_b = ((M + bm - 1) // bm, ld_sa), ((N + bn - 1) // bn, ld_sb)
        if pow2:
            sa = np.exp2(rng.integers(-3, 2, shape_a)).astype(np.float32)
            sb = np.exp2(rng.integers(-3, 2, shape_b)).astype(np.float32)
        else:
            sa = rng.uniform(0.0, 1.0, shape_a).astype(np.float32)
            sb = rng.uniform(0.0, 1.0, shape_b).astype(np.float32)
        out[f"{tag}_shape"] = np.array([M, N, K, bm, bn, bk, ld_sa, ld_sb, int(e_bf16)], np.int32)
        out[f"{tag}_a"], out[f"{tag}_q"], out[f"{tag}_sa"], out[f"{tag}_sb"] = a, q, sa, sb
        out[f"{tag}_e"] = ref.gemm_f8_i8_abscale(a, q, sa, sb, M, N, K, K, K, N, block, ld_sa, ld_sb, e_bf16=e_bf16)

    case("ab128_i4_bf16", 256, 256, 512, (128, 128, 128), -8, 7, True)
    case("ab128_i4_f16_pow2", 256, 512, 384, (128, 128, 128), -8, 7, False, pow2=True)
    case("ab_small_blocks_i8", 40, 48, 300, (16, 32, 128), -128, 127, False, pad=(1, 2))
    case("ab_ragged_i4", 33, 20, 200, (8, 4, 64), -8, 7, True, pad=(0, 3))
    return out


def main() -> None:
    ref = binding.load_ref()
    if ref is None:
        raise SystemExit("reference tree not mounted: cannot regenerate golden vectors")
    np.savez_compressed(os.path.join(OUT, "scalars.npz"), **scalars(ref))
    np.savez_compressed(os.path.join(OUT, "gemm_cases.npz"), **gemm_cases(ref))
    np.savez_compressed(os.path.join(OUT, "splitk_shapes.npz"), **splitk_shapes(ref))
    np.savez_compressed(os.path.join(OUT, "abscale_cases.npz"), **abscale_cases(ref))
    for f in ("scalars.npz", "gemm_cases.npz", "splitk_shapes.npz", "abscale_cases.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
