/*
 * TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement (CPU, gcc) of the host arithmetic that defines "correct" for the
 * weight-quantized GEMM path of Composable Kernel.  It is the checker the GPU kernels are
 * compared against; it is never linked into, loaded by or shipped with the product library, and
 * the product has no CPU fallback.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py may load the shared object built from this file.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against the reference's own
 * host templates (oracle/ref_shim.cpp -> oracle/_ref/libck_ref.so, built from /root/reference)
 * in tests/test_oracle_vs_ref.py, and against golden vectors generated from that library and
 * committed under tests/golden/ (tests/test_oracle_golden.py), which also run where the reference
 * is absent.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference root).
 * No floating-point contraction may be applied to this file (built with -ffp-contract=off): the
 * reference's host loops are separate multiply and add on x86-64.
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* bit helpers                                                                                 */
/* ------------------------------------------------------------------------------------------- */
static inline uint32_t f32_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float bits_f32(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* include/ck/utility/data_type.hpp:10-14 (half_t = _Float16) + type_convert.hpp:17-25:
 * float -> half is static_cast<_Float16>, i.e. IEEE binary16 round-to-nearest-even. */
uint16_t wqo_f32_to_f16(float f)
{
    const uint32_t x    = f32_bits(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax   = x & 0x7FFFFFFFu;
    if(ax >= 0x7F800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | (ax > 0x7F800000u ? (0x0200u | ((ax >> 13) & 0x3FFu)) : 0u));
    if(ax >= 0x477FF000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7C00u);
    if(ax < 0x33000001u) /* <= 2^-25 rounds to zero (ties-to-even at exactly 2^-25) */
        return (uint16_t)sign;
    int32_t e      = (int32_t)(ax >> 23) - 127; /* unbiased */
    uint32_t mant  = (ax & 0x7FFFFFu) | 0x800000u;
    uint32_t shift = 13;
    uint32_t hexp;
    if(e < -14)
    { /* subnormal half */
        shift += (uint32_t)(-14 - e);
        hexp = 0;
    }
    else
        hexp = (uint32_t)(e + 15);
    uint32_t q    = mant >> shift;
    uint32_t rem  = mant & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if(rem > half || (rem == half && (q & 1u)))
        q += 1;
    /* q carries the implicit bit for normals: (hexp-1)<<10 + q folds a mantissa carry upward */
    uint32_t h = hexp ? (((hexp - 1) << 10) + q) : q;
    return (uint16_t)(sign | h);
}

/* half -> float is exact (static_cast<float>(_Float16)). */
float wqo_f16_to_f32(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e    = (h >> 10) & 0x1Fu;
    uint32_t m          = h & 0x3FFu;
    if(e == 0)
    {
        if(m == 0)
            return bits_f32(sign);
        int sh = 0;
        while(!(m & 0x400u))
        {
            m <<= 1;
            ++sh;
        }
        m &= 0x3FFu;
        return bits_f32(sign | ((uint32_t)(113 - sh) << 23) | (m << 13));
    }
    if(e == 31)
        return bits_f32(sign | 0x7F800000u | (m << 13));
    return bits_f32(sign | ((e + 112u) << 23) | (m << 13));
}

/* include/ck/utility/type_convert.hpp:53-64: float -> bhalf_t keeps the upper 16 bits
 * (truncation, not round-to-nearest). */
uint16_t wqo_f32_to_bf16(float f) { return (uint16_t)(f32_bits(f) >> 16); }
/* include/ck/utility/type_convert.hpp:40-51 */
float wqo_bf16_to_f32(uint16_t b) { return bits_f32((uint32_t)b << 16); }

/* include/ck/utility/f8_utils.hpp:186-267 (run_cast_from_f8<f8_t,float,negative_zero_nan=true>),
 * reached from type_convert.hpp:383-396.  FNUZ E4M3: bias 8, 0x80 = NaN, no infinities. */
float wqo_f8_fnuz_to_f32(uint8_t x)
{
    if(x == 0)
        return 0.0f;
    if(x == 0x80)
        return bits_f32(0x7F800001u); /* NumericUtils<float>::NaN, data_type.hpp:1172 */
    const uint32_t sign = x >> 7;
    uint32_t mant       = x & 7u;
    int exponent        = (x & 0x7F) >> 3;
    if(exponent == 0)
    { /* subnormal input: normalise */
        int sh = 0;
        while(!(mant & 4u))
        {
            mant <<= 1;
            ++sh;
        }
        sh += 1;
        mant     = (mant << 1) & 7u;
        exponent = 1 - sh;
    }
    exponent += 127 - 8; /* exp_low_cutoff - 1 = 128 - 8 + 1 - 1 - 1 */
    return bits_f32((sign << 31) | ((uint32_t)exponent << 23) | (mant << 20));
}

/* type_convert.hpp:17-25: float -> int8_t is static_cast (toward zero). */
int8_t wqo_f32_to_i8(float x) { return (int8_t)x; }

/* ------------------------------------------------------------------------------------------- */
/* element-wise functors                                                                       */
/* ------------------------------------------------------------------------------------------- */

/* reference_fpAintB_gemm.hpp:110  v_converted_b = type_convert<half>(v_b) * v_scale
 * (_Float16 * _Float16: the product of two binary16 values is exact in binary32, so rounding it
 * once to binary16 is the correctly rounded half multiply). */
uint16_t wqo_dequant_f16(int8_t q, uint16_t scale)
{
    const float hq = wqo_f16_to_f32(wqo_f32_to_f16((float)q)); /* int8 -> half is exact */
    return wqo_f32_to_f16(hq * wqo_f16_to_f32(scale));
}

/* binary_element_wise_operation.hpp:182-190  Multiply::operator()<bhalf_t>(y, int8, bhalf):
 * y = type_convert<bhalf>(type_convert<float>(x0) * type_convert<float>(x1)) */
uint16_t wqo_multiply_bf16(int8_t q, uint16_t scale)
{
    return wqo_f32_to_bf16((float)q * wqo_bf16_to_f32(scale));
}

/* unary_element_wise_operation.hpp:443-452  host FastGelu<float,float>.  c1/c2 are formed in
 * double and narrowed; exp is the double-precision libm exp applied to the promoted float. */
float wqo_fastgelu(float x)
{
    const float c1  = (float)(-2.0 * 0.035677f);
    const float c2  = (float)(-2.0 * 0.797885f);
    const float u   = x * (c1 * x * x + c2);
    const float emu = (float)exp((double)u);
    return x / (1.f + emu);
}

/* quantization_operation.hpp:39-45  Activation_Mul_Clamp<PassThrough>(int8& y, const int32& x) */
int8_t wqo_act_mul_clamp(float requant, int32_t x)
{
    float y = (float)x;
    y       = requant * y;
    y       = y < -128.f ? -128.f : (y > 127.f ? 127.f : y); /* math::clamp = min(max(x,lo),hi) */
    return (int8_t)y;
}
/* reference_gemm.hpp:96-98 hands the float accumulator to the functor above; overload resolution
 * binds it to the (int8&, const int32&) form through a float -> int32 conversion (toward zero). */
int8_t wqo_act_mul_clamp_from_float(float requant, float acc)
{
    return wqo_act_mul_clamp(requant, (int32_t)acc);
}

/* quantization_operation.hpp:39-45 and :149-156 with Activation = PassThrough or Relu
 * (unary_element_wise_operation.hpp:409-419: y = x > 0 ? x : 0): the int32 bias is added in int32, the sum goes to
 * float, through the activation, times requantScale, clamped, and to int8 toward zero. */
int8_t wqo_add_act_mul_clamp(float requant, int32_t x, int32_t bias, int relu)
{
    float y = (float)(x + bias);
    if(relu)
        y = y > 0 ? y : 0;
    y = requant * y;
    y = y < -128.f ? -128.f : (y > 127.f ? 127.f : y);
    return (int8_t)y;
}

/* element_wise_operation.hpp:252-275  MultiplyMultiply: e = cvt(c * d0 * d1), left to right */
uint16_t wqo_multiply_multiply_f16(float c, float d0, float d1)
{
    const float x = c * d0 * d1;
    return wqo_f32_to_f16(x);
}
uint16_t wqo_multiply_multiply_bf16(float c, float d0, float d1)
{
    const float x = c * d0 * d1;
    return wqo_f32_to_bf16(x);
}
/* element_wise_operation.hpp:277-295  MultiplyAddFastGelu<bhalf,float,bhalf,bhalf> */
uint16_t wqo_multiply_add_fastgelu_bf16(float c, uint16_t d0, uint16_t d1)
{
    const float x0 = c * wqo_bf16_to_f32(d0) + wqo_bf16_to_f32(d1);
    return wqo_f32_to_bf16(wqo_fastgelu(x0));
}

/* element_wise_operation.hpp:200-231  MultiplyAdd<bhalf,float,bhalf,bhalf>: "type_convert<bhalf_t>(c) * d0 + d1" with
 * bhalf_t = unsigned short, i.e. INTEGER arithmetic on the three 16-bit patterns, truncated to 16 bits on assignment
 * (the low 16 bits do not depend on how the 32-bit product wraps).  Reproduced as written. */
uint16_t wqo_multiply_add_bf16(float c, uint16_t d0, uint16_t d1)
{
    return (uint16_t)((uint32_t)wqo_f32_to_bf16(c) * (uint32_t)d0 + (uint32_t)d1);
}

/* type_convert.hpp:507-549  bf16_convert_rtn<bhalf_t, float>, reached through the functor ConvertBF16RTN
 * (unary_element_wise_operation.hpp:236-251): round to nearest even on the upper 16 bits; Inf/NaN patterns are not
 * incremented, and a NaN whose payload lies only in the lower 16 bits keeps a mantissa bit (0x10000 is or-ed in). */
uint16_t wqo_bf16_convert_rtn(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    const int finite = (~u & 0x7F800000u) != 0;
    const int lownan = !finite && (u & 0xFFFFu);
    if(finite)
        u += 0x7FFFu + ((u >> 16) & 1u);
    if(lownan)
        u |= 0x10000u;
    return (uint16_t)(u >> 16);
}

/* ------------------------------------------------------------------------------------------- */
/* input generators                                                                            */
/* ------------------------------------------------------------------------------------------- */

/* std::mt19937 (32-bit Mersenne twister, the standard's parameters) */
typedef struct
{
    uint32_t mt[624];
    int idx;
} mt19937_t;

static void mt_seed(mt19937_t* g, uint32_t seed)
{
    g->mt[0] = seed;
    for(int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937_t* g)
{
    if(g->idx >= 624)
    {
        for(int i = 0; i < 624; ++i)
        {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7FFFFFFFu);
            g->mt[i]   = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9D2C5680u;
    y ^= (y << 15) & 0xEFC60000u;
    y ^= y >> 18;
    return y;
}
/* libstdc++ std::uniform_real_distribution<float>(a,b)(mt19937): one 32-bit draw,
 * generate_canonical<float,24>: float(draw) / 2^32 computed in float, values that round up to 1
 * are replaced by nextafter(1,0); result = canonical * (b - a) + a. */
static float uniform_real_f32(mt19937_t* g, float a, float b)
{
    float r = (float)mt_next(g) * (1.0f / 4294967296.0f);
    if(r >= 1.0f)
        r = nextafterf(1.0f, 0.0f);
    return r * (b - a) + a;
}

/* library/include/ck/library/utility/fill.hpp:18-41 (FillUniformDistribution) and :62-86
 * (FillUniformDistributionIntegerValue): a fresh mt19937(11939) per call,
 * uniform_real_distribution<float>(a,b), optional std::round, type_convert<T>.
 * dtype: 0 half, 1 bhalf, 2 int8, 3 float. */
void wqo_fill_uniform(void* out, size_t n, int dtype, float a, float b, int integer_value)
{
    mt19937_t g;
    mt_seed(&g, 11939u);
    for(size_t i = 0; i < n; ++i)
    {
        float v = uniform_real_f32(&g, a, b);
        if(integer_value)
            v = roundf(v);
        switch(dtype)
        {
        case 0: ((uint16_t*)out)[i] = wqo_f32_to_f16(v); break;
        case 1: ((uint16_t*)out)[i] = wqo_f32_to_bf16(v); break;
        case 2: ((int8_t*)out)[i] = (int8_t)v; break;
        default: ((float*)out)[i] = v; break;
        }
    }
}

/* host_tensor_generator.hpp:111-122 GeneratorTensor_2<int8_t> driven by
 * host_tensor.hpp:362-376 GenerateTensorValue(num_thread = 1): (i0,i1) visited row-major in
 * index space, value = rand() % (hi - lo) + lo from the C library's global stream. */
void wqo_generator2_i8(int8_t* out, size_t rows, size_t cols, size_t sr, size_t sc, int lo, int hi,
                       int seed)
{
    if(seed >= 0)
        srand((unsigned)seed);
    for(size_t i = 0; i < rows; ++i)
        for(size_t j = 0; j < cols; ++j)
            out[i * sr + j * sc] = (int8_t)((rand() % (hi - lo)) + lo);
}

/* ------------------------------------------------------------------------------------------- */
/* threading: library/include/ck/library/utility/host_tensor.hpp:187-245 ParallelTensorFunctor */
/* flat index space M*N split into num_thread contiguous chunks, one thread each               */
/* ------------------------------------------------------------------------------------------- */
typedef void (*mn_fn)(void* ctx, int m, int n);
typedef struct
{
    mn_fn f;
    void* ctx;
    size_t begin, end;
    int N;
} chunk_t;

static void* chunk_main(void* p)
{
    chunk_t* c = (chunk_t*)p;
    for(size_t i = c->begin; i < c->end; ++i)
        c->f(c->ctx, (int)(i / (size_t)c->N), (int)(i % (size_t)c->N));
    return NULL;
}

static void parallel_mn(mn_fn f, void* ctx, int M, int N, int nthreads)
{
    if(nthreads < 1)
        nthreads = 1;
    const size_t total = (size_t)M * (size_t)N;
    const size_t per   = (total + (size_t)nthreads - 1) / (size_t)nthreads;
    pthread_t* th      = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    chunk_t* ch        = (chunk_t*)malloc(sizeof(chunk_t) * (size_t)nthreads);
    int started        = 0;
    for(int t = 0; t < nthreads; ++t)
    {
        size_t b = (size_t)t * per, e = b + per;
        if(b >= total)
            break;
        if(e > total)
            e = total;
        ch[t] = (chunk_t){f, ctx, b, e, N};
        if(nthreads == 1)
            chunk_main(&ch[t]);
        else
            pthread_create(&th[t], NULL, chunk_main, &ch[t]);
        ++started;
    }
    if(nthreads > 1)
        for(int t = 0; t < started; ++t)
            pthread_join(th[t], NULL);
    free(th);
    free(ch);
}

/* ------------------------------------------------------------------------------------------- */
/* GEMMs                                                                                       */
/* ------------------------------------------------------------------------------------------- */

/* reference_fpAintB_gemm.hpp:62-127 with <half,int8,half,half,float,PT,PT,PT>:
 *   acc(float) += float(A(m,k)) * float( half(B(k,n)) * scale(n) ),  k ascending;
 *   C(m,n) = half(acc).
 * A (M,K) row-major lda; B int8 (K,N) column-major ldb; scale has lengths (K,N), strides (0,1);
 * C (M,N) row-major ldc.  k_begin/k_end select a K slice (split-K partials); when partial != NULL
 * the fp32 accumulator is stored there (row-major, ld = N) instead of the half result. */
typedef struct
{
    const uint16_t* a;
    const int8_t* b;
    const uint16_t* scale;
    uint16_t* c;
    float* partial;
    int N, lda, ldb, ldc, k_begin, k_end;
} fpaintb_ctx;

static void fpaintb_f16_mn(void* p, int m, int n)
{
    const fpaintb_ctx* x = (const fpaintb_ctx*)p;
    float acc            = 0.f;
    const float s        = wqo_f16_to_f32(x->scale[n]);
    for(int k = x->k_begin; k < x->k_end; ++k)
    {
        const float va = wqo_f16_to_f32(x->a[(size_t)m * x->lda + k]);
        const float hq = (float)x->b[(size_t)n * x->ldb + k]; /* exact in half */
        const float vb = wqo_f16_to_f32(wqo_f32_to_f16(hq * s));
        acc += va * vb;
    }
    if(x->partial)
        x->partial[(size_t)m * x->N + n] = acc;
    else
        x->c[(size_t)m * x->ldc + n] = wqo_f32_to_f16(acc);
}

void wqo_fpAintB_gemm_f16(const uint16_t* a, const int8_t* b, const uint16_t* scale, uint16_t* c,
                          int M, int N, int K, int lda, int ldb, int ldc, int nthreads)
{
    fpaintb_ctx x = {a, b, scale, c, NULL, N, lda, ldb, ldc, 0, K};
    parallel_mn(fpaintb_f16_mn, &x, M, N, nthreads);
}

void wqo_fpAintB_gemm_f16_partial(const uint16_t* a, const int8_t* b, const uint16_t* scale,
                                  float* partial, int M, int N, int lda, int ldb, int k_begin,
                                  int k_end, int nthreads)
{
    fpaintb_ctx x = {a, b, scale, NULL, partial, N, lda, ldb, 0, k_begin, k_end};
    parallel_mn(fpaintb_f16_mn, &x, M, N, nthreads);
}

/* The same reference loop (reference_fpAintB_gemm.hpp:62-127) with a scale that also depends on k: the reference
 * reads scale_k_n_(k, n) through a tensor descriptor, per-channel scales being the strides-(0,1) case.  Group-wise
 * scales are the tensor whose rows repeat group_k times: scale(k, n) = scale[(k / group_k) * ld_scale + n], i.e. one
 * row of N scales per group of group_k consecutive k (ceil(K / group_k) rows, the shape of the B1 scale tensor in
 * example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:126-152 with a block of one column). */
typedef struct
{
    const uint16_t* a;
    const int8_t* b;
    const uint16_t* scale;
    uint16_t* c;
    int K, lda, ldb, ldc, group_k, ld_scale;
} fpaintb_grouped_ctx;

static void fpaintb_f16_grouped_mn(void* p, int m, int n)
{
    const fpaintb_grouped_ctx* x = (const fpaintb_grouped_ctx*)p;
    float acc                    = 0.f;
    for(int k = 0; k < x->K; ++k)
    {
        const float s  = wqo_f16_to_f32(x->scale[(size_t)(k / x->group_k) * x->ld_scale + n]);
        const float va = wqo_f16_to_f32(x->a[(size_t)m * x->lda + k]);
        const float hq = (float)x->b[(size_t)n * x->ldb + k]; /* exact in half */
        const float vb = wqo_f16_to_f32(wqo_f32_to_f16(hq * s));
        acc += va * vb;
    }
    x->c[(size_t)m * x->ldc + n] = wqo_f32_to_f16(acc);
}

void wqo_fpAintB_gemm_f16_grouped(const uint16_t* a, const int8_t* b, const uint16_t* scale, uint16_t* c, int M,
                                  int N, int K, int lda, int ldb, int ldc, int group_k, int ld_scale, int nthreads)
{
    fpaintb_grouped_ctx x = {a, b, scale, c, K, lda, ldb, ldc, group_k, ld_scale};
    parallel_mn(fpaintb_f16_grouped_mn, &x, M, N, nthreads);
}

/* reference_gemm.hpp:60-108 with <int8,int8,int8,float,PT,PT,Activation_Mul_Clamp<PT>>
 * (example/14_gemm_quantization/gemm_xdl_quantization_int8.cpp:95-96): ComputeType = int8,
 * acc(float) += float(a) * float(b); e = Activation_Mul_Clamp(acc -> int32). */
typedef struct
{
    const int8_t *a, *b;
    int8_t* e;
    int K, lda, ldb, lde;
    float requant;
} i8_ctx;

static void i8_requant_mn(void* p, int m, int n)
{
    const i8_ctx* x = (const i8_ctx*)p;
    float acc       = 0.f;
    for(int k = 0; k < x->K; ++k)
        acc += (float)x->a[(size_t)m * x->lda + k] * (float)x->b[(size_t)n * x->ldb + k];
    x->e[(size_t)m * x->lde + n] = wqo_act_mul_clamp_from_float(x->requant, acc);
}

void wqo_gemm_i8_requant(const int8_t* a, const int8_t* b, int8_t* e, int M, int N, int K, int lda,
                         int ldb, int lde, float requant, int nthreads)
{
    i8_ctx x = {a, b, e, K, lda, ldb, lde, requant};
    parallel_mn(i8_requant_mn, &x, M, N, nthreads);
}

/* example/14_gemm_quantization/gemm_xdl_bias_relu_quantization_int8.cpp:97-103,211-229 verify path:
 * c = ReferenceGemm<int8,int8,int32,int32,PT,PT,PT> (int32 accumulation, exact), then per element
 * Add_Activation_Mul_Clamp<Act>(e, c(m,n), bias(n)); bias == NULL is Activation_Mul_Clamp<Act>(e, c(m,n)). */
typedef struct
{
    const int8_t *a, *b;
    const int32_t* bias;
    int8_t* e;
    int K, lda, ldb, lde, relu;
    float requant;
} i8b_ctx;

static void i8_bias_requant_mn(void* p, int m, int n)
{
    const i8b_ctx* x = (const i8b_ctx*)p;
    int32_t acc      = 0;
    for(int k = 0; k < x->K; ++k)
        acc += (int32_t)x->a[(size_t)m * x->lda + k] * (int32_t)x->b[(size_t)n * x->ldb + k];
    x->e[(size_t)m * x->lde + n] = wqo_add_act_mul_clamp(x->requant, acc, x->bias ? x->bias[n] : 0, x->relu);
}

void wqo_gemm_i8_bias_requant(const int8_t* a, const int8_t* b, const int32_t* bias, int8_t* e, int M, int N,
                              int K, int lda, int ldb, int lde, float requant, int relu, int nthreads)
{
    i8b_ctx x = {a, b, bias, e, K, lda, ldb, lde, relu, requant};
    parallel_mn(i8_bias_requant_mn, &x, M, N, nthreads);
}

/* example/60_gemm_multi_ABD/gemm_multi_ABD_xdl_fastgelu_bf16_i8.cpp:236-272 verify path:
 *   b(k,n) = Multiply(b0(k,n) int8, b1(n) bhalf)               (truncated to bhalf)
 *   c(m,n) = ReferenceGemm<bhalf,bhalf,float,float,PT,PT,PT>   (fp32, k ascending)
 *   e(m,n) = epilogue(c): 0 PassThrough, 1 FastGelu<bhalf,float>, 2 MultiplyAddFastGelu with
 *            per-column bhalf d0[n], d1[n]  (example/60 ..._multiply_bias_fastgelu_...:245-266),
 *            and the binary functors of the multi-ABD instance family with per-column bhalf d0[n]
 *            (library/include/ck/library/tensor_operation_instance/gpu/gemm_multi_abd.hpp:24-260):
 *            3 Add<bhalf>(float, bhalf)               binary_element_wise_operation.hpp:78-85
 *            4 AddFastGelu<bhalf,float,bhalf>         binary_element_wise_operation.hpp:511-522
 *            5 Multiply<bhalf>(float, bhalf)          binary_element_wise_operation.hpp:180-188
 *            6 MultiplyFastGelu<bhalf,float,bhalf>    binary_element_wise_operation.hpp:578-590
 *            7 MultiplyAdd<bhalf,float,bhalf,bhalf>   element_wise_operation.hpp:200-231 (d0, d1; bit-pattern arithmetic)
 *            8 ConvertBF16RTN<bhalf,float>            unary_element_wise_operation.hpp:236-251 (float -> bhalf, nearest even;
 *              the C element-wise operation of example/01_gemm/gemm_xdl_bf16_rtn.cpp:21)
 *   b1 == NULL: the "multiply" instances, B = PassThrough(b0) (int8 -> bhalf, exact), scale in the epilogue.
 * B element (k,n) at k*b_sk + n*b_sn.  D element (m,n) at m*ldd + n: the examples' d_m_n(m, n) over a row-major
 * descriptor with row stride StrideD (gemm_multi_ABD_xdl_bias_fastgelu_bf16_i8.cpp:86,140,263), 0 = one value per
 * column. */
typedef struct
{
    const uint16_t* a;
    const int8_t* b0;
    const uint16_t* b1;
    uint16_t* e;
    const uint16_t *d0, *d1;
    int K, lda, b_sk, b_sn, lde, epilogue;
    int ldd0, ldd1;
    int group_k, ld_b1; /* group_k > 0: b1 holds one row of scales per group_k consecutive k, rows ld_b1 apart */
} bf16_ctx;

static void bf16_i8_mn(void* p, int m, int n)
{
    const bf16_ctx* x = (const bf16_ctx*)p;
    float acc         = 0.f;
    for(int k = 0; k < x->K; ++k)
    {
        const float va = wqo_bf16_to_f32(x->a[(size_t)m * x->lda + k]);
        const int8_t q = x->b0[(size_t)k * x->b_sk + (size_t)n * x->b_sn];
        const uint16_t* b1 = x->b1 && x->group_k > 0 ? x->b1 + (size_t)(k / x->group_k) * x->ld_b1 : x->b1;
        const float vb     = b1 ? wqo_bf16_to_f32(wqo_multiply_bf16(q, b1[n])) : (float)q;
        acc += va * vb;
    }
    const uint16_t* d0 = x->d0 ? x->d0 + (size_t)m * x->ldd0 : 0;
    const uint16_t* d1 = x->d1 ? x->d1 + (size_t)m * x->ldd1 : 0;
    uint16_t y;
    if(x->epilogue == 0)
        y = wqo_f32_to_bf16(acc);
    else if(x->epilogue == 1)
        y = wqo_f32_to_bf16(wqo_fastgelu(acc));
    else if(x->epilogue == 2)
        y = wqo_multiply_add_fastgelu_bf16(acc, d0[n], d1[n]);
    else if(x->epilogue == 3)
        y = wqo_f32_to_bf16(acc + wqo_bf16_to_f32(d0[n]));
    else if(x->epilogue == 4)
        y = wqo_f32_to_bf16(wqo_fastgelu(acc + wqo_bf16_to_f32(d0[n])));
    else if(x->epilogue == 5)
        y = wqo_f32_to_bf16(acc * wqo_bf16_to_f32(d0[n]));
    else if(x->epilogue == 7)
        y = wqo_multiply_add_bf16(acc, d0[n], d1[n]);
    else if(x->epilogue == 8)
        y = wqo_bf16_convert_rtn(acc);
    else
        y = wqo_f32_to_bf16(wqo_fastgelu(acc * wqo_bf16_to_f32(d0[n])));
    x->e[(size_t)m * x->lde + n] = y;
}

void wqo_gemm_bf16_i8(const uint16_t* a, const int8_t* b0, const uint16_t* b1, uint16_t* e, int M,
                      int N, int K, int lda, int b_sk, int b_sn, int lde, int epilogue,
                      const uint16_t* d0, const uint16_t* d1, int nthreads)
{
    bf16_ctx x = {a, b0, b1, e, d0, d1, K, lda, b_sk, b_sn, lde, epilogue, 0, 0, 0, 0};
    parallel_mn(bf16_i8_mn, &x, M, N, nthreads);
}

/* The same verify path with a B1 that varies along k: the Multiply functor (binary_element_wise_operation.hpp:182-190) sees
 * b1(k, n) = scale row k / group_k (one row of N bhalf scales per group_k consecutive k, rows ld_b1 apart) -- the (K, N)
 * B1 tensor of gemm_multi_ABD_xdl_fastgelu_bf16_i8.cpp:137-139 with rows that repeat per group instead of a stride of 0. */
void wqo_gemm_bf16_i8_grouped(const uint16_t* a, const int8_t* b0, const uint16_t* b1, uint16_t* e, int M, int N, int K,
                              int lda, int b_sk, int b_sn, int lde, int epilogue, const uint16_t* d0, const uint16_t* d1,
                              int group_k, int ld_b1, int nthreads)
{
    bf16_ctx x = {a, b0, b1, e, d0, d1, K, lda, b_sk, b_sn, lde, epilogue, 0, 0, group_k, ld_b1};
    parallel_mn(bf16_i8_mn, &x, M, N, nthreads);
}

/* Same, with D tensors addressed as full row-major (M, N) matrices of row stride ldd0 / ldd1 (0 = per column). */
void wqo_gemm_bf16_i8_dmat(const uint16_t* a, const int8_t* b0, const uint16_t* b1, uint16_t* e, int M,
                           int N, int K, int lda, int b_sk, int b_sn, int lde, int epilogue,
                           const uint16_t* d0, int ldd0, const uint16_t* d1, int ldd1, int nthreads)
{
    bf16_ctx x = {a, b0, b1, e, d0, d1, K, lda, b_sk, b_sn, lde, epilogue, ldd0, ldd1, 0, 0};
    parallel_mn(bf16_i8_mn, &x, M, N, nthreads);
}

/* example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8.cpp:262-300 verify path with
 * an int8-held int4 B: c = ReferenceGemm<f8,int8,float,float,PT,PT,PT>;
 * e = half(c*d0[n]*d1[m]) (MultiplyMultiply) or, fastgelu != 0, half(FastGelu(c*d0[n]*d1[m])). */
typedef struct
{
    const uint8_t* a;
    const int8_t* b;
    const float *d0, *d1;
    uint16_t* e;
    int K, lda, ldb, lde, fastgelu;
} f8_ctx;

static void f8_i8_mm_mn(void* p, int m, int n)
{
    const f8_ctx* x = (const f8_ctx*)p;
    float acc       = 0.f;
    for(int k = 0; k < x->K; ++k)
        acc += wqo_f8_fnuz_to_f32(x->a[(size_t)m * x->lda + k]) *
               (float)x->b[(size_t)n * x->ldb + k];
    const float v = acc * x->d0[n] * x->d1[m];
    x->e[(size_t)m * x->lde + n] =
        x->fastgelu ? wqo_f32_to_f16(wqo_fastgelu(v)) : wqo_f32_to_f16(v);
}

void wqo_gemm_f8_i8_mm(const uint8_t* a, const int8_t* b, const float* d0, const float* d1,
                       uint16_t* e, int M, int N, int K, int lda, int ldb, int lde, int fastgelu,
                       int nthreads)
{
    f8_ctx x = {a, b, d0, d1, e, K, lda, ldb, lde, fastgelu};
    parallel_mn(f8_i8_mm_mn, &x, M, N, nthreads);
}

/* example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:262-283 (host verification of
 * DeviceGemmMultiD_ABScale): both operands are dequantised to fp32 first --
 *   a'(m,k) = float(a0(m,k)) * a1(m / bm, k / bk)      a1 row-major, ld_sa    (A0Layout = RowMajor descriptor, :126-133)
 *   b'(k,n) = float(b0(k,n)) * b1(k / bk, n / bn)      b1 column-major: (kb, nb) at kb + nb * ld_sb (B0Layout = ColumnMajor, :135-139)
 * -- then ReferenceGemm<float, float, float, float> (fp32 multiply and add, k ascending) and the conversion of the sum to
 * the output type (CDEElementOp = PassThrough, :50-54).  Here b0 holds int4 / int8 weights in int8 (K contiguous per
 * column) instead of the example's fp8 bytes; e_bf16 != 0: bf16 output (the example's EDataType; truncating convert),
 * else fp16.  e4m3fn != 0: the activation bytes are OCP E4M3 instead of the reference's FNUZ. */
typedef struct
{
    const uint8_t* a;
    const int8_t* b;
    const float* sa;
    const float* sb;
    uint16_t* e;
    int K, lda, ldb, lde, bm, bn, bk, ld_sa, ld_sb, e_bf16, e4m3fn;
} f8_ab_ctx;

static float wqo_f8_e4m3fn_to_f32_local(uint8_t x)
{
    if((x & 0x7F) == 0x7F)
        return NAN;
    const int e = (x >> 3) & 15, m = x & 7;
    const float v = e == 0 ? (float)m * 0.001953125f : ldexpf((float)(8 + m), e - 10);
    return (x & 0x80) ? -v : v;
}

static void f8_ab_mn(void* p, int m, int n)
{
    const f8_ab_ctx* x = (const f8_ab_ctx*)p;
    float acc          = 0.f;
    for(int k = 0; k < x->K; ++k)
    {
        const uint8_t byte = x->a[(size_t)m * x->lda + k];
        const float a0     = x->e4m3fn ? wqo_f8_e4m3fn_to_f32_local(byte) : wqo_f8_fnuz_to_f32(byte);
        const float av     = a0 * x->sa[(size_t)(m / x->bm) * x->ld_sa + k / x->bk];
        const float bv     = (float)x->b[(size_t)n * x->ldb + k] * x->sb[(size_t)(k / x->bk) + (size_t)(n / x->bn) * x->ld_sb];
        acc += av * bv;
    }
    x->e[(size_t)m * x->lde + n] = x->e_bf16 ? wqo_f32_to_bf16(acc) : wqo_f32_to_f16(acc);
}

void wqo_gemm_f8_i8_abscale(const uint8_t* a, const int8_t* b, const float* sa, const float* sb, uint16_t* e, int M, int N, int K,
                            int lda, int ldb, int lde, int bm, int bn, int bk, int ld_sa, int ld_sb, int e_bf16, int e4m3fn,
                            int nthreads)
{
    f8_ab_ctx x = {a, b, sa, sb, e, K, lda, ldb, lde, bm, bn, bk, ld_sa, ld_sb, e_bf16, e4m3fn};
    parallel_mn(f8_ab_mn, &x, M, N, nthreads);
}

/* ------------------------------------------------------------------------------------------- */
/* split-K arithmetic                                                                          */
/* ------------------------------------------------------------------------------------------- */

/* include/ck/tensor_operation/gpu/grid/gridwise_gemm_xdl_cshuffle_v3.hpp:192-197 CalculateKRead:
 * KRead = ceil(K / (KBatch * v)) * v, v = the K vector granule; slice z covers
 * [z*KRead, min(K,(z+1)*KRead)) (:583-590: the last batch takes the remainder). */
int wqo_splitk_kread(int K, int kbatch, int granule)
{
    const int per = kbatch * granule;
    return (K + per - 1) / per * granule;
}

/* device_gemm_xdl_cshuffle_v3r1.hpp:209-270 RunReduce: sum fp32 partials [KBatch][M][N] over dim 0
 * in ascending batch order, then convert (half). */
void wqo_splitk_reduce_f16(const float* partials, int kbatch, int M, int N, uint16_t* c, int ldc)
{
    for(int m = 0; m < M; ++m)
        for(int n = 0; n < N; ++n)
        {
            float s = 0.f;
            for(int z = 0; z < kbatch; ++z)
                s += partials[((size_t)z * M + m) * N + n];
            c[(size_t)m * ldc + n] = wqo_f32_to_f16(s);
        }
}

/* ------------------------------------------------------------------------------------------- */
/* packed int4 weight format (defined by this build; the reference has no packed int4:         */
/* include/ck/utility/data_type.hpp:10-14, example/64_fpAintB_gemm/run_gemm_example.inc:8-10)  */
/* ------------------------------------------------------------------------------------------- */
/* Values are the reference's signed int4 range -8..7 held in int8 (K,N) column-major, ldq.
 * Stored offset-binary (nibble = q + 8, mirroring example/64_fpAintB_gemm/common.hpp:66-74's
 * "+128" preprocess), column n at byte offset n*ld_bytes, K/2 bytes per column.  Within each
 * group of 8 consecutive k (one little-endian 32-bit word) nibble p (bits 4p..4p+3) holds
 *   k = 8*g + 2*p          for p = 0..3   (even k in the low half-word)
 *   k = 8*g + 2*(p-4) + 1  for p = 4..7   (odd  k in the high half-word)
 * so that (w >> 4i) & 0x000F000F yields the pair (k=2i, k=2i+1) in the two fp16 lanes.
 * Returns 0 on success, -1 if K is not a multiple of 8 or a value is outside -8..7. */
int wqo_pack_int4(const int8_t* q, int ldq, int K, int N, uint8_t* out, int ld_bytes)
{
    if(K % 8 != 0)
        return -1;
    for(int n = 0; n < N; ++n)
        for(int g = 0; g < K / 8; ++g)
        {
            uint32_t w = 0;
            for(int p = 0; p < 8; ++p)
            {
                const int k = 8 * g + (p < 4 ? 2 * p : 2 * (p - 4) + 1);
                const int v = q[(size_t)n * ldq + k];
                if(v < -8 || v > 7)
                    return -1;
                w |= (uint32_t)(v + 8) << (4 * p);
            }
            uint8_t* dst = out + (size_t)n * ld_bytes + 4 * g;
            dst[0]       = (uint8_t)(w & 0xFF);
            dst[1]       = (uint8_t)((w >> 8) & 0xFF);
            dst[2]       = (uint8_t)((w >> 16) & 0xFF);
            dst[3]       = (uint8_t)(w >> 24);
        }
    return 0;
}

/* The pre-shuffled order of the packed int4 weights (this engine's offline format for its 256-tile kernels; the
 * reference tree has no int4 counterpart).  Written from the format's definition, element by element: the nibble of
 * (k, n) keeps its position inside its 16-byte unit (the 32 consecutive k from 32 * (k / 32) of column n), and the unit
 * moves to
 *   (n / 32) * (ldk * 16)  +  (k / 64) * 1024  +  (32 * ((n % 32) / 16) + 16 * ((k % 64) / 32) + n % 16) * 16
 * bytes from the start (ldk = K pitch in elements, a multiple of 64).  Returns 0, or -1 on bad sizes. */
int wqo_preshuffle_int4(const uint8_t* packed, int ld_bytes, int K, int N, uint8_t* out, int ldk)
{
    if(K % 64 != 0 || N % 32 != 0 || ldk % 64 != 0 || ldk < K)
        return -1;
    for(int n = 0; n < N; ++n)
        for(int k = 0; k < K; k += 2) /* one byte holds two nibbles; byte index inside the unit is kept */
        {
            const int byte_in_col  = k / 8 * 4 + (k % 8) / 2; /* little-endian word k / 8, byte (k % 8) / 2 of it */
            const int byte_in_unit = byte_in_col % 16;
            const size_t dst       = (size_t)(n / 32) * ((size_t)ldk * 16) + (size_t)(k / 64) * 1024 +
                               (size_t)(32 * ((n % 32) / 16) + 16 * ((k % 64) / 32) + n % 16) * 16 + byte_in_unit;
            out[dst] = packed[(size_t)n * ld_bytes + byte_in_col];
        }
    return 0;
}

/* The pre-shuffled order of 8-bit weights, byte by byte from the format's definition: byte (k, n) keeps its position
 * k % 16 inside its 16-byte unit, and the unit moves to
 *   (n / 16) * (ldk * 16)  +  (k / 64) * 1024  +  (4 * (n % 16) + (k % 64) / 16) * 16. */
int wqo_preshuffle_int8(const int8_t* b, int ldb, int K, int N, int8_t* out, int ldk)
{
    if(K % 64 != 0 || N % 16 != 0 || ldk % 64 != 0 || ldk < K)
        return -1;
    for(int n = 0; n < N; ++n)
        for(int k = 0; k < K; ++k)
            out[(size_t)(n / 16) * ((size_t)ldk * 16) + (size_t)(k / 64) * 1024 + (size_t)(4 * (n % 16) + (k % 64) / 16) * 16 +
                k % 16] = b[(size_t)n * ldb + k];
    return 0;
}

int wqo_unpack_int4(const uint8_t* in, int ld_bytes, int K, int N, int8_t* q, int ldq)
{
    if(K % 8 != 0)
        return -1;
    for(int n = 0; n < N; ++n)
        for(int g = 0; g < K / 8; ++g)
        {
            const uint8_t* src = in + (size_t)n * ld_bytes + 4 * g;
            const uint32_t w   = (uint32_t)src[0] | ((uint32_t)src[1] << 8) |
                               ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
            for(int p = 0; p < 8; ++p)
            {
                const int k               = 8 * g + (p < 4 ? 2 * p : 2 * (p - 4) + 1);
                q[(size_t)n * ldq + k] = (int8_t)((int)((w >> (4 * p)) & 0xF) - 8);
            }
        }
    return 0;
}

/* example/64_fpAintB_gemm/common.hpp:51-83 UnsignedWeightPreprocessor<int8_t>: u8 = int8 + 128 */
void wqo_offset_u8(const int8_t* q, uint8_t* out, size_t n)
{
    for(size_t i = 0; i < n; ++i)
        out[i] = (uint8_t)((uint8_t)q[i] + 128u);
}
