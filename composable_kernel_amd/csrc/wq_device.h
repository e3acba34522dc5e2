// Device-side building blocks shared by the gfx950 weight-quantized GEMM kernels:
// vector types, in-register dequantisation of B fragments, epilogue functors and stores.
//
// Numerics follow Composable Kernel's host reference (paths relative to its tree):
//   fp16 dequant  half(q) * scale, rounded once to half   reference_fpAintB_gemm.hpp:110
//   bf16 dequant  trunc_bf16(float(q) * float(scale))     binary_element_wise_operation.hpp:182-190
//   float -> bf16 truncation                              include/ck/utility/type_convert.hpp:53-64
//   float -> half round-to-nearest-even                   include/ck/utility/type_convert.hpp:17-25
// This file must be compiled with -ffp-contract=off: the reference's epilogues are separate
// multiplies and adds.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace wq {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef int i4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));

enum AKind
{
    A_F16  = 0,
    A_BF16 = 1
};
enum BKind
{
    B_I4P  = 0, // packed offset-binary int4, 8 k per 32-bit word
    B_U8O  = 1, // uint8 = q + 128
    B_I8   = 2  // signed int8
};

template <int BK>
struct BInfo;
template <>
struct BInfo<B_I4P>
{
    static constexpr int kBytesPer8K = 4; // bytes holding 8 consecutive k of one column
};
template <>
struct BInfo<B_U8O>
{
    static constexpr int kBytesPer8K = 8;
};
template <>
struct BInfo<B_I8>
{
    static constexpr int kBytesPer8K = 8;
};

__device__ __forceinline__ h2_t as_h2(uint32_t u) { return __builtin_bit_cast(h2_t, u); }
__device__ __forceinline__ uint32_t as_u32(h2_t h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return as_f32((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_trunc(float f) { return (uint16_t)(as_u32(f) >> 16); }

// ---------------------------------------------------------------------------------------------
// Per-lane scale operand.  For fp16 activations it is {s,s} as a packed half pair; for bf16
// activations it is s widened to fp32 plus the precomputed -128*s (exact: power-of-two multiple).
// ---------------------------------------------------------------------------------------------
template <int AK>
struct ScaleReg;
template <>
struct ScaleReg<A_F16>
{
    h2_t s2;
    __device__ __forceinline__ void set(uint16_t bits)
    {
        const uint32_t u = (uint32_t)bits | ((uint32_t)bits << 16);
        s2               = as_h2(u);
    }
};
template <>
struct ScaleReg<A_BF16>
{
    float s, m128s;
    __device__ __forceinline__ void set(uint16_t bits)
    {
        s     = bf16_to_f32(bits);
        m128s = -128.0f * s;
    }
};

// 1.0 in the activation type: the weight scale when the caller passes none (scales applied by the epilogue)
template <int AK>
__device__ __forceinline__ constexpr uint16_t unit_scale_bits()
{
    return AK == A_F16 ? 0x3C00 : 0x3F80;
}

// ---------------------------------------------------------------------------------------------
// Dequantise the 8 consecutive-k weights one lane feeds to one MFMA (k = 8g .. 8g+7 of one
// column) into the 4 VGPRs of a 16-bit operand fragment, element j <-> k = 8g + j.
// ---------------------------------------------------------------------------------------------

// fp16 x packed int4.  0x6400 | n is the half 1024 + n, so (w & 0x000F000F) | 0x64006400 holds
// the pair (1024 + nibble_i, 1024 + nibble_{i+4}); subtracting 1032 leaves q = n - 8 exactly.
// Nibbles at bit 4 are kept in place, giving 1024 + 16 n, and (x * 1/16 - 72) = n - 8 exactly.
// The multiply by the scale is then the single rounding the reference performs.
__device__ __forceinline__ u4_t dequant8(const ScaleReg<A_F16>& sc, uint32_t w,
                                         std::integral_constant<int, B_I4P>)
{
    const h2_t k1032 = {(_Float16)1032.0f, (_Float16)1032.0f};
    const h2_t k16th = {(_Float16)0.0625f, (_Float16)0.0625f};
    const h2_t km72  = {(_Float16)-72.0f, (_Float16)-72.0f};
    const uint32_t w8 = w >> 8;
    const h2_t t0     = as_h2((w & 0x000F000Fu) | 0x64006400u);
    const h2_t t1     = as_h2((w & 0x00F000F0u) | 0x64006400u);
    const h2_t t2     = as_h2((w8 & 0x000F000Fu) | 0x64006400u);
    const h2_t t3     = as_h2((w8 & 0x00F000F0u) | 0x64006400u);
    const h2_t q0     = t0 - k1032;
    const h2_t q1     = __builtin_elementwise_fma(t1, k16th, km72);
    const h2_t q2     = t2 - k1032;
    const h2_t q3     = __builtin_elementwise_fma(t3, k16th, km72);
    u4_t r;
    r[0] = as_u32(q0 * sc.s2);
    r[1] = as_u32(q1 * sc.s2);
    r[2] = as_u32(q2 * sc.s2);
    r[3] = as_u32(q3 * sc.s2);
    return r;
}

// fp16 x 8-bit weights.  v_perm_b32 builds the half pair (0x64 b1, 0x64 b0) = (1024 + b1,
// 1024 + b0); subtracting 1152 leaves b - 128 = q exactly (the u8 -> half conversion of
// unary_element_wise_operation.hpp:1060-1091).  Signed bytes are first flipped to offset form.
template <int BKIND>
__device__ __forceinline__ u4_t dequant8_bytes_f16(const ScaleReg<A_F16>& sc, u2_t w)
{
    if constexpr(BKIND == B_I8)
    {
        w[0] ^= 0x80808080u;
        w[1] ^= 0x80808080u;
    }
    const h2_t k1152 = {(_Float16)1152.0f, (_Float16)1152.0f};
    const h2_t t0    = as_h2(__builtin_amdgcn_perm(0x64646464u, w[0], 0x04010400u));
    const h2_t t1    = as_h2(__builtin_amdgcn_perm(0x64646464u, w[0], 0x04030402u));
    const h2_t t2    = as_h2(__builtin_amdgcn_perm(0x64646464u, w[1], 0x04010400u));
    const h2_t t3    = as_h2(__builtin_amdgcn_perm(0x64646464u, w[1], 0x04030402u));
    u4_t r;
    r[0] = as_u32((t0 - k1152) * sc.s2);
    r[1] = as_u32((t1 - k1152) * sc.s2);
    r[2] = as_u32((t2 - k1152) * sc.s2);
    r[3] = as_u32((t3 - k1152) * sc.s2);
    return r;
}

// bf16 x 8-bit weights.  float(q) * float(scale) has at most 16 significant bits, so it is exact
// in fp32 and fma(float(q + 128), s, -128 s) returns exactly that product; the reference then
// truncates to bf16, which is "keep the upper half-word" (v_perm_b32 packs two of them).
__device__ __forceinline__ float u8_to_f32(uint32_t w, int byte)
{
    return (float)((w >> (8 * byte)) & 0xFFu); // v_cvt_f32_ubyteN
}
template <int BKIND>
__device__ __forceinline__ u4_t dequant8_bytes_bf16(const ScaleReg<A_BF16>& sc, u2_t w)
{
    if constexpr(BKIND == B_I8)
    {
        w[0] ^= 0x80808080u;
        w[1] ^= 0x80808080u;
    }
    // two products per v_pk_fma_f32
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t s2 = {sc.s, sc.s}, m2 = {sc.m128s, sc.m128s};
    u4_t r;
#pragma unroll
    for(int d = 0; d < 2; ++d)
    {
        const f2_t u01 = {u8_to_f32(w[d], 0), u8_to_f32(w[d], 1)};
        const f2_t u23 = {u8_to_f32(w[d], 2), u8_to_f32(w[d], 3)};
        const f2_t p01 = __builtin_elementwise_fma(u01, s2, m2);
        const f2_t p23 = __builtin_elementwise_fma(u23, s2, m2);
        r[2 * d]     = __builtin_amdgcn_perm(as_u32(p01[1]), as_u32(p01[0]), 0x07060302u);
        r[2 * d + 1] = __builtin_amdgcn_perm(as_u32(p23[1]), as_u32(p23[0]), 0x07060302u);
    }
    return r;
}

// bf16 x packed int4: same exactness argument with float(n) * s - 8 s.
__device__ __forceinline__ u4_t dequant8(const ScaleReg<A_BF16>& sc, uint32_t w,
                                         std::integral_constant<int, B_I4P>)
{
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const float m8s = sc.m128s * 0.0625f; // -8 s, exact
    const f2_t s2 = {sc.s, sc.s}, m2 = {m8s, m8s};
    u4_t r;
#pragma unroll
    for(int i = 0; i < 4; ++i)
    {
        const f2_t n = {(float)((w >> (4 * i)) & 0xFu), (float)((w >> (4 * i + 16)) & 0xFu)};
        const f2_t v = __builtin_elementwise_fma(n, s2, m2); // one v_pk_fma_f32 for the pair
        r[i]         = __builtin_amdgcn_perm(as_u32(v[1]), as_u32(v[0]), 0x07060302u);
    }
    return r;
}

// Scalar dequantisation of one weight (generic kernel); same arithmetic as above.
template <int AK>
__device__ __forceinline__ float dequant_scalar(int q, uint16_t scale_bits)
{
    if constexpr(AK == A_F16)
    {
        const _Float16 s = __builtin_bit_cast(_Float16, scale_bits);
        const _Float16 v = (_Float16)(float)q * s; // one rounding to half
        return (float)v;
    }
    else
    {
        const float p = (float)q * bf16_to_f32(scale_bits);
        return as_f32(as_u32(p) & 0xFFFF0000u);
    }
}

// ---------------------------------------------------------------------------------------------
// Epilogues.  x is the fp32 accumulator of output element (m, n).
// ---------------------------------------------------------------------------------------------
struct EpilogueArgs
{
    int kind;      // wq_epilogue
    int c_dtype;   // wq_dtype of the output
    float requant; // Activation_Mul_Clamp
    const void* d0;
    const void* d1;
    // Element (m, n) of a D operand.  16-bit kinds (2, 6 - 9): both row-major, d[m * ldd + n].  MultiplyMultiply kinds
    // (3, 4): d0 row-major d0[m * ldd0 + n], d1 column-major d1[m + n * ldd1].  ldd = 0 is the broadcast form
    // (one value per column, or per row for the column-major d1).
    int ldd0, ldd1;
};

// FastGelu, unary_element_wise_operation.hpp:443-465: u = x (c1 x^2 + c2), y = x / (1 + exp(u)) with
// c1 = -2 * 0.035677, c2 = -2 * 0.797885.  Evaluated as the reference's device form does (hardware exponential and
// reciprocal instead of the host's libm exp and true division), with log2(e) folded into the two constants so that the
// exponential is one v_exp_f32: u' = x fma(c1', x^2, c2'), y = x rcp(1 + exp2(u')).  The scalar and the two-wide
// forms perform the same IEEE operations, so every kernel and the split-K reduce produce the same bits.
typedef float f2_t __attribute__((ext_vector_type(2)));
constexpr float kGeluC1L2E = (float)(-2.0 * 0.035677 * 1.44269504088896341);
constexpr float kGeluC2L2E = (float)(-2.0 * 0.797885 * 1.44269504088896341);

__device__ __forceinline__ float fast_gelu(float x)
{
    const float inner = __builtin_fmaf(kGeluC1L2E, x * x, kGeluC2L2E);
    const float emu   = __builtin_amdgcn_exp2f(x * inner); // (inf for large arguments gives rcp(inf) = 0 as it should)
    return x * __builtin_amdgcn_rcpf(1.f + emu);
}

// two outputs per instruction where the ISA has packed fp32 arithmetic (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32)
__device__ __forceinline__ f2_t fast_gelu2(f2_t x)
{
    const f2_t c1 = {kGeluC1L2E, kGeluC1L2E}, c2 = {kGeluC2L2E, kGeluC2L2E}, one = {1.f, 1.f};
    const f2_t u  = x * __builtin_elementwise_fma(c1, x * x, c2);
    const f2_t d  = one + f2_t{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
    return x * f2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// Flat index of element (i, j) of a D operand with leading dimension ld, in 64 bits: full (M, N) D tensors of problems
// whose operands exceed 2 GiB have indices beyond INT_MAX.
__device__ __forceinline__ size_t d_at(int i, int ld, int j) { return (size_t)i * (size_t)ld + (size_t)j; }

__device__ __forceinline__ float load_d_as_f32(const void* d, size_t idx, int c_dtype)
{
    // MultiplyAddFastGelu operands have the output's 16-bit type
    const uint16_t bits = ((const uint16_t*)d)[idx];
    if(c_dtype == 1 /*WQ_BF16*/)
        return bf16_to_f32(bits);
    return (float)__builtin_bit_cast(_Float16, bits);
}

// MultiplyAdd<bhalf_t, float, bhalf_t, bhalf_t> (element_wise_operation.hpp:200-231): integer arithmetic on the 16-bit
// patterns, low 16 bits kept.  Operands and result travel as the floats whose upper halves are those patterns, so the
// truncating float -> bf16 store that follows returns the result bits unchanged.
__device__ __forceinline__ float multiply_add_bf16_bits(float c, float d0, float d1)
{
    const uint32_t cb = __builtin_bit_cast(uint32_t, c) >> 16, a = __builtin_bit_cast(uint32_t, d0) >> 16,
                   b  = __builtin_bit_cast(uint32_t, d1) >> 16;
    return __builtin_bit_cast(float, (cb * a + b) << 16);
}

// ConvertBF16RTN (unary_element_wise_operation.hpp:236-251 -> type_convert.hpp:507-549): round to nearest even on the
// upper 16 bits; Inf / NaN patterns are not incremented, a NaN whose payload sits in the lower half keeps a mantissa
// bit.  Returned as the float whose upper half is the result, for the truncating float -> bf16 store that follows.
__device__ __forceinline__ float convert_bf16_rtn_bits(float x)
{
    uint32_t u        = __builtin_bit_cast(uint32_t, x);
    const bool finite = (~u & 0x7F800000u) != 0;
    u                 = finite ? u + 0x7FFFu + ((u >> 16) & 1u) : ((u & 0xFFFFu) ? (u | 0x10000u) : u);
    return __builtin_bit_cast(float, u & 0xFFFF0000u);
}

__device__ __forceinline__ float apply_epilogue(const EpilogueArgs& e, float x, int m, int n)
{
    switch(e.kind)
    {
    case 1: return fast_gelu(x);
    case 2: {
        // element_wise_operation.hpp:277-295: FastGelu(c * d0 + d1), separate multiply and add
        const float t = x * load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype);
        return fast_gelu(t + load_d_as_f32(e.d1, d_at(m, e.ldd1, n), e.c_dtype));
    }
    case 3: {
        // element_wise_operation.hpp:252-275: c * d0 * d1 left to right
        const float t = x * ((const float*)e.d0)[d_at(m, e.ldd0, n)];
        return t * ((const float*)e.d1)[d_at(n, e.ldd1, m)];
    }
    case 4: {
        const float t = x * ((const float*)e.d0)[d_at(m, e.ldd0, n)];
        return fast_gelu(t * ((const float*)e.d1)[d_at(n, e.ldd1, m)]);
    }
    // binary_element_wise_operation.hpp: Add :78-85, AddFastGelu :511-522, Multiply (float x bhalf), MultiplyFastGelu
    // :578-590 -- the second operand is d0[n] of the output's 16-bit type
    case 6: return x + load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype);
    case 7: return fast_gelu(x + load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype));
    case 8: return x * load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype);
    case 9: return fast_gelu(x * load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype));
    case 13:
        return multiply_add_bf16_bits(x, load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype), load_d_as_f32(e.d1, d_at(m, e.ldd1, n), e.c_dtype));
    case 14: return convert_bf16_rtn_bits(x);
    default: return x;
    }
}

// Same functors with the kind fixed at compile time (keeps unrolled epilogues small).
template <int KIND>
__device__ __forceinline__ float apply_epilogue_k(const EpilogueArgs& e, float x, int m, int n)
{
    if constexpr(KIND == 1)
        return fast_gelu(x);
    else if constexpr(KIND == 2)
    {
        const float t = x * load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype);
        return fast_gelu(t + load_d_as_f32(e.d1, d_at(m, e.ldd1, n), e.c_dtype));
    }
    else if constexpr(KIND == 3)
    {
        const float t = x * ((const float*)e.d0)[d_at(m, e.ldd0, n)];
        return t * ((const float*)e.d1)[d_at(n, e.ldd1, m)];
    }
    else if constexpr(KIND == 4)
    {
        const float t = x * ((const float*)e.d0)[d_at(m, e.ldd0, n)];
        return fast_gelu(t * ((const float*)e.d1)[d_at(n, e.ldd1, m)]);
    }
    else if constexpr(KIND == 6)
        return x + load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype);
    else if constexpr(KIND == 7)
        return fast_gelu(x + load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype));
    else if constexpr(KIND == 8)
        return x * load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype);
    else if constexpr(KIND == 9)
        return fast_gelu(x * load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype));
    else if constexpr(KIND == 13)
        return multiply_add_bf16_bits(x, load_d_as_f32(e.d0, d_at(m, e.ldd0, n), e.c_dtype), load_d_as_f32(e.d1, d_at(m, e.ldd1, n), e.c_dtype));
    else if constexpr(KIND == 14)
        return convert_bf16_rtn_bits(x);
    else
        return x;
}

// The same functors on the four consecutive columns n .. n+3 of row m that one accumulator fragment holds, two
// columns per instruction.  Operation order per element is that of apply_epilogue_k (identical bits).
template <int KIND>
__device__ __forceinline__ f4_t apply_epilogue4(const EpilogueArgs& e, f4_t v, int m, int n)
{
    if constexpr(KIND == 13 || KIND == 14)
    {
        f4_t y;
#pragma unroll
        for(int j = 0; j < 4; ++j)
            y[j] = apply_epilogue_k<KIND>(e, v[j], m, n + j);
        return y;
    }
    f2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
    if constexpr(KIND == 2 || (KIND >= 6 && KIND <= 9))
    {
        const size_t i0 = d_at(m, e.ldd0, n);
        const f2_t d0l = {load_d_as_f32(e.d0, i0, e.c_dtype), load_d_as_f32(e.d0, i0 + 1, e.c_dtype)};
        const f2_t d0h = {load_d_as_f32(e.d0, i0 + 2, e.c_dtype), load_d_as_f32(e.d0, i0 + 3, e.c_dtype)};
        if constexpr(KIND == 6 || KIND == 7)
            lo = lo + d0l, hi = hi + d0h;
        else
            lo = lo * d0l, hi = hi * d0h;
        if constexpr(KIND == 2)
        {
            const size_t i1 = d_at(m, e.ldd1, n);
            lo = lo + f2_t{load_d_as_f32(e.d1, i1, e.c_dtype), load_d_as_f32(e.d1, i1 + 1, e.c_dtype)};
            hi = hi + f2_t{load_d_as_f32(e.d1, i1 + 2, e.c_dtype), load_d_as_f32(e.d1, i1 + 3, e.c_dtype)};
        }
    }
    else if constexpr(KIND == 3 || KIND == 4)
    {
        const float* d0 = (const float*)e.d0 + d_at(m, e.ldd0, n);
        const float* d1 = (const float*)e.d1 + d_at(n, e.ldd1, m);
        lo = lo * f2_t{d0[0], d0[1]}, hi = hi * f2_t{d0[2], d0[3]};
        lo = lo * f2_t{d1[0], d1[(size_t)e.ldd1]}, hi = hi * f2_t{d1[2 * (size_t)e.ldd1], d1[3 * (size_t)e.ldd1]};
    }
    if constexpr(KIND == 1 || KIND == 2 || KIND == 4 || KIND == 7 || KIND == 9)
        lo = fast_gelu2(lo), hi = fast_gelu2(hi);
    return f4_t{lo[0], lo[1], hi[0], hi[1]};
}

// The broadcast forms of the D operands (ldd0 == ldd1 == 0) depend on the column only -- or, for MultiplyMultiply's
// d1, on the row only -- so a tile epilogue can fetch them once per column block / row block instead of once per
// accumulator fragment.  load_d0_cols4 / load_d1_cols4: the four values of columns n .. n+3 as fp32; load_d1_row: the
// row's value.  apply_epilogue4_bcast performs the operations of apply_epilogue4 in the same order on them.
template <int KIND>
__device__ __forceinline__ f4_t load_d0_cols4(const EpilogueArgs& e, int n)
{
    if constexpr(KIND == 3 || KIND == 4)
    {
        const float* d = (const float*)e.d0 + n;
        return f4_t{d[0], d[1], d[2], d[3]};
    }
    else if constexpr(KIND == 2 || KIND == 13 || (KIND >= 6 && KIND <= 9))
        return f4_t{load_d_as_f32(e.d0, n, e.c_dtype), load_d_as_f32(e.d0, n + 1, e.c_dtype),
                    load_d_as_f32(e.d0, n + 2, e.c_dtype), load_d_as_f32(e.d0, n + 3, e.c_dtype)};
    else
        return f4_t{0.f, 0.f, 0.f, 0.f};
}

template <int KIND>
__device__ __forceinline__ f4_t load_d1_cols4(const EpilogueArgs& e, int n)
{
    if constexpr(KIND == 2 || KIND == 13)
        return f4_t{load_d_as_f32(e.d1, n, e.c_dtype), load_d_as_f32(e.d1, n + 1, e.c_dtype),
                    load_d_as_f32(e.d1, n + 2, e.c_dtype), load_d_as_f32(e.d1, n + 3, e.c_dtype)};
    else
        return f4_t{0.f, 0.f, 0.f, 0.f};
}

template <int KIND>
__device__ __forceinline__ float load_d1_row(const EpilogueArgs& e, int m)
{
    if constexpr(KIND == 3 || KIND == 4)
        return ((const float*)e.d1)[m];
    else
        return 0.f;
}

template <int KIND>
__device__ __forceinline__ f4_t apply_epilogue4_bcast(f4_t v, f4_t d0, f4_t d1n, float d1m)
{
    if constexpr(KIND == 13)
        return f4_t{multiply_add_bf16_bits(v[0], d0[0], d1n[0]), multiply_add_bf16_bits(v[1], d0[1], d1n[1]),
                    multiply_add_bf16_bits(v[2], d0[2], d1n[2]), multiply_add_bf16_bits(v[3], d0[3], d1n[3])};
    if constexpr(KIND == 14)
        return f4_t{convert_bf16_rtn_bits(v[0]), convert_bf16_rtn_bits(v[1]), convert_bf16_rtn_bits(v[2]),
                    convert_bf16_rtn_bits(v[3])};
    f2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
    if constexpr(KIND == 6 || KIND == 7)
        lo = lo + f2_t{d0[0], d0[1]}, hi = hi + f2_t{d0[2], d0[3]};
    else if constexpr(KIND == 2 || KIND == 3 || KIND == 4 || KIND == 8 || KIND == 9)
        lo = lo * f2_t{d0[0], d0[1]}, hi = hi * f2_t{d0[2], d0[3]};
    if constexpr(KIND == 2)
        lo = lo + f2_t{d1n[0], d1n[1]}, hi = hi + f2_t{d1n[2], d1n[3]};
    if constexpr(KIND == 3 || KIND == 4)
        lo = lo * f2_t{d1m, d1m}, hi = hi * f2_t{d1m, d1m};
    if constexpr(KIND == 1 || KIND == 2 || KIND == 4 || KIND == 7 || KIND == 9)
        lo = fast_gelu2(lo), hi = fast_gelu2(hi);
    return f4_t{lo[0], lo[1], hi[0], hi[1]};
}

// four fp32 values -> four 16-bit outputs in two registers (fp16: round to nearest even; bf16: truncation)
template <bool BF16>
__device__ __forceinline__ u2_t pack_out4(f4_t y)
{
    u2_t pk;
    if constexpr(BF16)
    {
        pk[0] = __builtin_amdgcn_perm(as_u32(y[1]), as_u32(y[0]), 0x07060302u);
        pk[1] = __builtin_amdgcn_perm(as_u32(y[3]), as_u32(y[2]), 0x07060302u);
    }
    else
    {
        const h2_t a = __builtin_convertvector(f2_t{y[0], y[1]}, h2_t), b = __builtin_convertvector(f2_t{y[2], y[3]}, h2_t);
        pk[0] = as_u32(a), pk[1] = as_u32(b);
    }
    return pk;
}

// Store one accumulator fragment: lane holds C[m][n .. n+3] (fp32) of a row-major C.
template <int KIND, bool BF16>
__device__ __forceinline__ void store_c4(const EpilogueArgs& e, void* c, int ldc, f4_t v, int m, int n)
{
    *(u2_t*)((uint16_t*)c + (size_t)m * ldc + n) = pack_out4<BF16>(apply_epilogue4<KIND>(e, v, m, n));
}

// The requantisation family on an int32 accumulator: kind 5 Activation_Mul_Clamp<PassThrough>, 10 <Relu>,
// 11 / 12 Add_Activation_Mul_Clamp<PassThrough / Relu> with an int32 bias per column
// (quantization_operation.hpp:39-45,149-156: the bias is added in int32, then float, activation, scale, clamp)
__device__ __forceinline__ int8_t requant_epilogue(int kind, float requant, int acc, const int32_t* bias, int n)
{
    if(kind >= 11)
        acc += bias[n];
    float y = (float)acc;
    if(kind == 10 || kind == 12)
        y = y > 0.f ? y : 0.f;
    y = fminf(fmaxf(requant * y, -128.f), 127.f);
    return (int8_t)y;
}

// The same family on one accumulator fragment (four consecutive columns of a row), packed into four bytes, with the
// per-column work hoisted by the caller: bias4 = the columns' bias (zeros for kinds without one), relu_floor = 0 for the
// Relu kinds and -infinity otherwise (max(y, 0) is the reference's y > 0 ? y : 0 on converted integers).
__device__ __forceinline__ uint32_t requant_pack4(i4_t acc, i4_t bias4, float relu_floor, float requant)
{
    uint32_t pk = 0;
#pragma unroll
    for(int j = 0; j < 4; ++j)
    {
        float y = fmaxf((float)(acc[j] + bias4[j]), relu_floor);
        y       = fminf(fmaxf(requant * y, -128.f), 127.f);
        pk |= (uint32_t)(uint8_t)(int8_t)y << (8 * j);
    }
    return pk;
}

__device__ __forceinline__ float requant_relu_floor(int kind) { return kind == 10 || kind == 12 ? 0.f : -__builtin_inff(); }

__device__ __forceinline__ uint16_t cvt_out16(float v, int c_dtype)
{
    if(c_dtype == 1 /*WQ_BF16*/)
        return f32_to_bf16_trunc(v);
    return __builtin_bit_cast(uint16_t, (_Float16)v);
}

__device__ __forceinline__ float f8_fnuz_to_f32(uint8_t x)
{
    // E4M3 FNUZ: bias 8, 0x80 = NaN, subnormals m * 2^-10 (f8_utils.hpp:186-267)
    if(x == 0x80)
        return __builtin_nanf("");
    const int e   = (x >> 3) & 15;
    const int m   = x & 7;
    const float v = e == 0 ? (float)m * 0.0009765625f : __builtin_ldexpf((float)(8 + m), e - 11);
    return (x & 0x80) ? -v : v;
}

__device__ __forceinline__ float f8_e4m3fn_to_f32(uint8_t x)
{
    // OCP E4M3 ("e4m3fn"): bias 7, S.1111.111 = NaN, no infinities, subnormals m * 2^-9
    if((x & 0x7F) == 0x7F)
        return __builtin_nanf("");
    const int e   = (x >> 3) & 15;
    const int m   = x & 7;
    const float v = e == 0 ? (float)m * 0.001953125f : __builtin_ldexpf((float)(8 + m), e - 10);
    return (x & 0x80) ? -v : v;
}


} // namespace wq
