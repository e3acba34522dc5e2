// TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product library.
//
// A thin extern "C" layer over the reference's own host code, compiled from the headers where
// they lie under /root/reference (see oracle/Makefile, target _ref/libck_ref.so).  It exists to
//   (i)  pin oracle/wq_oracle.c (the plain-C restatement) bit-for-bit against the real reference,
//   (ii) generate the golden vectors committed under tests/golden/, and
//   (iii) serve as the "reference" CPU baseline that bench.py times beside the GPU number.
// Nothing here re-implements arithmetic: every function forwards to a reference template.
//
// Reference entry points used (paths relative to the reference root):
//   library/include/ck/library/reference_tensor_operation/cpu/reference_fpAintB_gemm.hpp:62-127
//   library/include/ck/library/reference_tensor_operation/cpu/reference_gemm.hpp:60-108
//   include/ck/utility/type_convert.hpp:17-64,383-396
//   include/ck/tensor_operation/gpu/element/{unary,binary,element_wise,quantization}_operation.hpp
//   library/include/ck/library/utility/{fill,host_tensor,host_tensor_generator}.hpp

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/library/utility/host_tensor.hpp"
#include "ck/library/utility/host_tensor_generator.hpp"
#include "ck/library/utility/fill.hpp"
#include "ck/library/reference_tensor_operation/cpu/reference_gemm.hpp"
#include "ck/library/reference_tensor_operation/cpu/reference_fpAintB_gemm.hpp"

namespace {

using ck::bhalf_t;
using ck::f8_t;
using ck::half_t;
namespace ew = ck::tensor_operation::element_wise;
using PT     = ew::PassThrough;

template <typename T, typename U>
T bits_to(U u)
{
    static_assert(sizeof(T) == sizeof(U));
    T t;
    std::memcpy(&t, &u, sizeof(T));
    return t;
}

HostTensorDescriptor desc2(std::size_t r, std::size_t c, std::size_t sr, std::size_t sc)
{
    return HostTensorDescriptor({r, c}, {sr, sc});
}

// Copy a strided user matrix into a reference Tensor with the same strides (storage copied
// verbatim, so element (i,j) lives at i*sr + j*sc in both).
template <typename T>
Tensor<T> load2(const void* p, std::size_t r, std::size_t c, std::size_t sr, std::size_t sc)
{
    Tensor<T> t(desc2(r, c, sr, sc));
    std::memcpy(t.mData.data(), p, t.mData.size() * sizeof(T));
    return t;
}

template <typename T>
void store2(const Tensor<T>& t, void* p)
{
    std::memcpy(p, t.mData.data(), t.mData.size() * sizeof(T));
}

} // namespace

extern "C" {

// ---- scalar conversions (type_convert.hpp) ------------------------------------------------
uint16_t ckref_f32_to_f16(float x) { return bits_to<uint16_t>(ck::type_convert<half_t>(x)); }
float ckref_f16_to_f32(uint16_t h) { return ck::type_convert<float>(bits_to<half_t>(h)); }
uint16_t ckref_f32_to_bf16(float x) { return ck::type_convert<bhalf_t>(x); }
float ckref_bf16_to_f32(uint16_t b) { return ck::type_convert<float>(bhalf_t(b)); }
float ckref_f8_to_f32(uint8_t v) { return ck::type_convert<float>(bits_to<f8_t>(v)); }
int8_t ckref_f32_to_i8(float x) { return ck::type_convert<int8_t>(x); }

// ---- element-wise functors -----------------------------------------------------------------
// reference_fpAintB_gemm.hpp:110  type_convert<half>(q) * scale
uint16_t ckref_dequant_f16(int8_t q, uint16_t scale)
{
    half_t v = ck::type_convert<half_t>(q) * bits_to<half_t>(scale);
    return bits_to<uint16_t>(v);
}
// binary_element_wise_operation.hpp:182-190  Multiply<bhalf>(int8, bhalf)
uint16_t ckref_multiply_bf16(int8_t q, uint16_t scale)
{
    bhalf_t y;
    ew::Multiply{}(y, q, bhalf_t(scale));
    return y;
}
// unary_element_wise_operation.hpp:443-452 host FastGelu<float,float>
float ckref_fastgelu(float x)
{
    float y;
    ew::FastGelu{}.template operator()<float, float>(y, x);
    return y;
}
// quantization_operation.hpp:39-45, reached from reference_gemm.hpp:96-98 with a float argument
int8_t ckref_act_mul_clamp_from_float(float requant, float acc)
{
    int8_t y = 0;
    ew::Activation_Mul_Clamp<PT>{requant, PT{}}(y, acc);
    return y;
}
int8_t ckref_act_mul_clamp(float requant, int32_t acc)
{
    int8_t y = 0;
    ew::Activation_Mul_Clamp<PT>{requant, PT{}}(y, acc);
    return y;
}
// element_wise_operation.hpp:252-275
uint16_t ckref_multiply_multiply_f16(float c, float d0, float d1)
{
    half_t e;
    ew::MultiplyMultiply{}(e, c, d0, d1);
    return bits_to<uint16_t>(e);
}
uint16_t ckref_multiply_multiply_bf16(float c, float d0, float d1)
{
    bhalf_t e;
    ew::MultiplyMultiply{}(e, c, d0, d1);
    return e;
}
// element_wise_operation.hpp:277-295
uint16_t ckref_multiply_add_fastgelu_bf16(float c, uint16_t d0, uint16_t d1)
{
    bhalf_t e;
    ew::MultiplyAddFastGelu{}(e, c, bhalf_t(d0), bhalf_t(d1));
    return e;
}
// element_wise_operation.hpp:200-231 (the bhalf_t specialisation)
uint16_t ckref_multiply_add_bf16(float c, uint16_t d0, uint16_t d1)
{
    bhalf_t e;
    ew::MultiplyAdd{}(e, c, bhalf_t(d0), bhalf_t(d1));
    return e;
}

// unary_element_wise_operation.hpp:236-251 -> type_convert.hpp:507-549
uint16_t ckref_bf16_convert_rtn(float x)
{
    bhalf_t e;
    ew::ConvertBF16RTN{}(e, x);
    return e;
}

// ---- input generators (fill.hpp:18-86, host_tensor_generator.hpp:84-169) -------------------
// dtype: 0 = half, 1 = bhalf, 2 = int8, 3 = float
void ckref_fill_uniform(void* out, size_t n, int dtype, float a, float b, int integer_value)
{
    auto run = [&](auto* p) {
        using T = std::remove_pointer_t<decltype(p)>;
        if(integer_value)
            ck::utils::FillUniformDistributionIntegerValue<T>{a, b}(p, p + n);
        else
            ck::utils::FillUniformDistribution<T>{a, b}(p, p + n);
    };
    switch(dtype)
    {
    case 0: run(static_cast<half_t*>(out)); break;
    case 1: run(static_cast<bhalf_t*>(out)); break;
    case 2: run(static_cast<int8_t*>(out)); break;
    default: run(static_cast<float*>(out)); break;
    }
}

// GeneratorTensor_2<int8_t>{lo,hi} over a (rows, cols) tensor with the given strides, drawn in
// (i0,i1) index order by GenerateTensorValue(num_thread = 1); srand(seed) first when seed >= 0.
void ckref_generator2_i8(int8_t* out, size_t rows, size_t cols, size_t sr, size_t sc, int lo,
                         int hi, int seed)
{
    if(seed >= 0)
        std::srand(static_cast<unsigned>(seed));
    Tensor<int8_t> t(desc2(rows, cols, sr, sc));
    t.GenerateTensorValue(GeneratorTensor_2<int8_t>{lo, hi});
    store2(t, out);
}

// ---- GEMMs ---------------------------------------------------------------------------------
// ReferencefpAintBGemm<half,int8,half,half,float,PT,PT,PT>; A (M,K) row-major lda; B int8 (K,N)
// column-major ldb; scale N values (lengths (K,N), strides (0,1)); C (M,N) row-major ldc.
void ckref_fpAintB_gemm_f16(const uint16_t* a, const int8_t* b, const uint16_t* scale,
                            uint16_t* c, int M, int N, int K, int lda, int ldb, int ldc)
{
    using Ref = ck::tensor_operation::host::
        ReferencefpAintBGemm<half_t, int8_t, half_t, half_t, float, PT, PT, PT>;
    auto ta = load2<half_t>(a, M, K, lda, 1);
    auto tb = load2<int8_t>(b, K, N, 1, ldb);
    auto ts = load2<half_t>(scale, K, N, 0, 1);
    Tensor<half_t> tc(desc2(M, N, ldc, 1));
    auto ref = Ref{};
    auto inv = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, ts, tc, PT{}, PT{}, PT{}));
    store2(tc, c);
}

// The same ReferencefpAintBGemm with a scale tensor that varies along k: lengths (K,N), strides (N,1), row k holding
// the N scales of group k / group_k (group-wise scales: scale[(k / group_k) * ld_scale + n]).
void ckref_fpAintB_gemm_f16_grouped(const uint16_t* a, const int8_t* b, const uint16_t* scale, uint16_t* c, int M,
                                    int N, int K, int lda, int ldb, int ldc, int group_k, int ld_scale)
{
    using Ref = ck::tensor_operation::host::
        ReferencefpAintBGemm<half_t, int8_t, half_t, half_t, float, PT, PT, PT>;
    auto ta = load2<half_t>(a, M, K, lda, 1);
    auto tb = load2<int8_t>(b, K, N, 1, ldb);
    Tensor<half_t> ts(desc2(K, N, N, 1));
    for(int k = 0; k < K; ++k)
        for(int n = 0; n < N; ++n)
            ts(k, n) = bits_to<half_t>(scale[(size_t)(k / group_k) * ld_scale + n]);
    Tensor<half_t> tc(desc2(M, N, ldc, 1));
    auto ref = Ref{};
    auto inv = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, ts, tc, PT{}, PT{}, PT{}));
    store2(tc, c);
}

// ReferenceGemm<int8,int8,int8,float,PT,PT,Activation_Mul_Clamp<PT>> (example/14 verify path).
// A row-major lda, B column-major ldb, E row-major lde.
void ckref_gemm_i8_requant(const int8_t* a, const int8_t* b, int8_t* e, int M, int N, int K,
                           int lda, int ldb, int lde, float requant)
{
    using Op  = ew::Activation_Mul_Clamp<PT>;
    using Ref = ck::tensor_operation::host::ReferenceGemm<int8_t, int8_t, int8_t, float, PT, PT, Op>;
    auto ta   = load2<int8_t>(a, M, K, lda, 1);
    auto tb   = load2<int8_t>(b, K, N, 1, ldb);
    Tensor<int8_t> te(desc2(M, N, lde, 1));
    auto ref = Ref{};
    auto inv = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, te, PT{}, PT{}, Op{requant, PT{}}));
    store2(te, e);
}

// example/14 bias + relu verify path (gemm_xdl_bias_relu_quantization_int8.cpp:97-103,211-229):
// ReferenceGemm<int8,int8,int32,int32,PT,PT,PT>, then Add_Activation_Mul_Clamp<Act>(e, c, bias(n)) per element;
// bias == nullptr applies Activation_Mul_Clamp<Act>(e, c).  relu selects Act = Relu, else PassThrough.
void ckref_gemm_i8_bias_requant(const int8_t* a, const int8_t* b, const int32_t* bias, int8_t* e, int M, int N,
                                int K, int lda, int ldb, int lde, float requant, int relu)
{
    using Ref = ck::tensor_operation::host::ReferenceGemm<int8_t, int8_t, int32_t, int32_t, PT, PT, PT>;
    auto ta   = load2<int8_t>(a, M, K, lda, 1);
    auto tb   = load2<int8_t>(b, K, N, 1, ldb);
    Tensor<int32_t> tc(desc2(M, N, N, 1));
    auto ref = Ref{};
    auto inv = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, tc, PT{}, PT{}, PT{}));
    for(int m = 0; m < M; ++m)
        for(int n = 0; n < N; ++n)
        {
            int8_t y = 0;
            if(bias && relu)
                ew::Add_Activation_Mul_Clamp<ew::Relu>{requant, ew::Relu{}}(y, tc(m, n), bias[n]);
            else if(bias)
                ew::Add_Activation_Mul_Clamp<PT>{requant, PT{}}(y, tc(m, n), bias[n]);
            else if(relu)
                ew::Activation_Mul_Clamp<ew::Relu>{requant, ew::Relu{}}(y, tc(m, n));
            else
                ew::Activation_Mul_Clamp<PT>{requant, PT{}}(y, tc(m, n));
            e[(size_t)m * lde + n] = y;
        }
}

// example/60 verify path: b = Multiply(b0 int8, b1 bhalf scale[n]); c = ReferenceGemm<bhalf,
// bhalf, float, float, PT,PT,PT>; e = epilogue(c).  B element (k,n) at k*b_sk + n*b_sn.
// epilogue: 0 PassThrough (float->bhalf), 1 FastGelu<bhalf,float>,
//           2 MultiplyAddFastGelu with d0[n], d1[n] bhalf (stride-0 rows),
//           3 Add, 4 AddFastGelu, 5 Multiply, 6 MultiplyFastGelu with d0[n] bhalf (the multi-ABD instance family),
//           7 MultiplyAdd<bhalf> (bit patterns), 8 ConvertBF16RTN (float -> bhalf, nearest even).
// b1 == nullptr: B = PassThrough(b0) (the "multiply" instances).
void ckref_gemm_bf16_i8(const uint16_t* a, const int8_t* b0, const uint16_t* b1, uint16_t* e,
                        int M, int N, int K, int lda, int b_sk, int b_sn, int lde, int epilogue,
                        const uint16_t* d0, const uint16_t* d1)
{
    auto ta  = load2<bhalf_t>(a, M, K, lda, 1);
    auto tb0 = load2<int8_t>(b0, K, N, b_sk, b_sn);
    Tensor<bhalf_t> tb(desc2(K, N, b_sk, b_sn));
    for(int n = 0; n < N; ++n)
        for(int k = 0; k < K; ++k)
        {
            if(b1)
                ew::Multiply{}(tb(k, n), tb0(k, n), bhalf_t(b1[n]));
            else
                PT{}(tb(k, n), tb0(k, n));
        }
    Tensor<float> tc(desc2(M, N, N, 1));
    using Ref =
        ck::tensor_operation::host::ReferenceGemm<bhalf_t, bhalf_t, float, float, PT, PT, PT>;
    auto ref = Ref{};
    auto inv = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, tc, PT{}, PT{}, PT{}));
    for(int m = 0; m < M; ++m)
        for(int n = 0; n < N; ++n)
        {
            bhalf_t y;
            if(epilogue == 0)
                PT{}(y, tc(m, n));
            else if(epilogue == 1)
                ew::FastGelu{}(y, tc(m, n));
            else if(epilogue == 2)
                ew::MultiplyAddFastGelu{}(y, tc(m, n), bhalf_t(d0[n]), bhalf_t(d1[n]));
            else if(epilogue == 3)
                ew::Add{}(y, tc(m, n), bhalf_t(d0[n]));
            else if(epilogue == 4)
                ew::AddFastGelu{}(y, tc(m, n), bhalf_t(d0[n]));
            else if(epilogue == 5)
                ew::Multiply{}(y, tc(m, n), bhalf_t(d0[n]));
            else if(epilogue == 7)
                ew::MultiplyAdd{}(y, tc(m, n), bhalf_t(d0[n]), bhalf_t(d1[n]));
            else if(epilogue == 8)
                ew::ConvertBF16RTN{}(y, tc(m, n));
            else
                ew::MultiplyFastGelu{}(y, tc(m, n), bhalf_t(d0[n]));
            e[static_cast<size_t>(m) * lde + n] = y;
        }
}

// The same path with a B1 tensor of lengths (K, N) and strides (N, 1) whose row k holds the scales of group k / group_k
// (rows of the caller's (K / group_k, ld_b1) array repeated): Multiply on B element by element, then ReferenceGemm.
void ckref_gemm_bf16_i8_grouped(const uint16_t* a, const int8_t* b0, const uint16_t* b1, uint16_t* e, int M, int N, int K,
                                int lda, int b_sk, int b_sn, int lde, int epilogue, const uint16_t* d0, const uint16_t* d1,
                                int group_k, int ld_b1)
{
    auto ta  = load2<bhalf_t>(a, M, K, lda, 1);
    auto tb0 = load2<int8_t>(b0, K, N, b_sk, b_sn);
    Tensor<bhalf_t> tb1(desc2(K, N, N, 1));
    for(int k = 0; k < K; ++k)
        for(int n = 0; n < N; ++n)
            tb1(k, n) = bhalf_t(b1[(size_t)(k / group_k) * ld_b1 + n]);
    Tensor<bhalf_t> tb(desc2(K, N, b_sk, b_sn));
    for(int n = 0; n < N; ++n)
        for(int k = 0; k < K; ++k)
            ew::Multiply{}(tb(k, n), tb0(k, n), tb1(k, n));
    Tensor<float> tc(desc2(M, N, N, 1));
    using Ref = ck::tensor_operation::host::ReferenceGemm<bhalf_t, bhalf_t, float, float, PT, PT, PT>;
    auto ref = Ref{};
    auto inv = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, tc, PT{}, PT{}, PT{}));
    for(int m = 0; m < M; ++m)
        for(int n = 0; n < N; ++n)
        {
            bhalf_t y;
            if(epilogue == 0)
                PT{}(y, tc(m, n));
            else if(epilogue == 1)
                ew::FastGelu{}(y, tc(m, n));
            else if(epilogue == 3)
                ew::Add{}(y, tc(m, n), bhalf_t(d0[n]));
            else if(epilogue == 4)
                ew::AddFastGelu{}(y, tc(m, n), bhalf_t(d0[n]));
            else
                ew::ConvertBF16RTN{}(y, tc(m, n));
            (void)d1;
            e[static_cast<size_t>(m) * lde + n] = y;
        }
}

// Same path with the D tensors addressed as the examples' d_m_n(m, n) over a row-major (M, N) descriptor of row
// stride ldd (example/60_gemm_multi_ABD/gemm_multi_ABD_xdl_bias_fastgelu_bf16_i8.cpp:86,140,263); 0 = per column.
void ckref_gemm_bf16_i8_dmat(const uint16_t* a, const int8_t* b0, const uint16_t* b1, uint16_t* e,
                             int M, int N, int K, int lda, int b_sk, int b_sn, int lde, int epilogue,
                             const uint16_t* d0, int ldd0, const uint16_t* d1, int ldd1)
{
    auto ta  = load2<bhalf_t>(a, M, K, lda, 1);
    auto tb0 = load2<int8_t>(b0, K, N, b_sk, b_sn);
    Tensor<bhalf_t> tb(desc2(K, N, b_sk, b_sn));
    for(int n = 0; n < N; ++n)
        for(int k = 0; k < K; ++k)
        {
            if(b1)
                ew::Multiply{}(tb(k, n), tb0(k, n), bhalf_t(b1[n]));
            else
                PT{}(tb(k, n), tb0(k, n));
        }
    Tensor<float> tc(desc2(M, N, N, 1));
    using Ref = ck::tensor_operation::host::ReferenceGemm<bhalf_t, bhalf_t, float, float, PT, PT, PT>;
    auto ref = Ref{};
    auto inv = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, tc, PT{}, PT{}, PT{}));
    Tensor<bhalf_t> td0(desc2(M, N, ldd0, 1)), td1(desc2(M, N, ldd1, 1));
    if(d0)
        for(size_t i = 0; i < td0.mData.size(); ++i)
            td0.mData[i] = bhalf_t(d0[i]);
    if(d1)
        for(size_t i = 0; i < td1.mData.size(); ++i)
            td1.mData[i] = bhalf_t(d1[i]);
    for(int m = 0; m < M; ++m)
        for(int n = 0; n < N; ++n)
        {
            bhalf_t y;
            if(epilogue == 2)
                ew::MultiplyAddFastGelu{}(y, tc(m, n), td0(m, n), td1(m, n));
            else if(epilogue == 3)
                ew::Add{}(y, tc(m, n), td0(m, n));
            else if(epilogue == 4)
                ew::AddFastGelu{}(y, tc(m, n), td0(m, n));
            else if(epilogue == 5)
                ew::Multiply{}(y, tc(m, n), td0(m, n));
            else if(epilogue == 7)
                ew::MultiplyAdd{}(y, tc(m, n), td0(m, n), td1(m, n));
            else
                ew::MultiplyFastGelu{}(y, tc(m, n), td0(m, n));
            e[static_cast<size_t>(m) * lde + n] = y;
        }
}

// example/35_splitK_gemm/gemm_xdl_splitk_reduce_bf16A_i8B.cpp:55-61 + run_gemm_splitk_reduce_multi_d_example.inc:260-266
// verify path: ReferenceGemm<bhalf, int8, bhalf, float, PT, PT, PT> straight on the int8 B (no scale), C bhalf.
void ckref_gemm_bf16_i8_unscaled(const uint16_t* a, const int8_t* b, uint16_t* c, int M, int N, int K, int lda,
                                 int b_sk, int b_sn, int ldc)
{
    auto ta = load2<bhalf_t>(a, M, K, lda, 1);
    auto tb = load2<int8_t>(b, K, N, b_sk, b_sn);
    Tensor<bhalf_t> tc(desc2(M, N, ldc, 1));
    using Ref = ck::tensor_operation::host::ReferenceGemm<bhalf_t, int8_t, bhalf_t, float, PT, PT, PT>;
    auto ref = Ref{};
    auto inv = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, tc, PT{}, PT{}, PT{}));
    for(int m = 0; m < M; ++m)
        for(int n = 0; n < N; ++n)
            c[static_cast<size_t>(m) * ldc + n] = tc(m, n);
}

// example/65 verify path with an int8-held int4 B: c = ReferenceGemm<f8,int8,float,float>;
// e = half(c * d0[n] * d1[m]) (MultiplyMultiply), optionally followed by FastGelu<half,float>
// applied to the fp32 product (fastgelu != 0: e = half(FastGelu(c*d0*d1))).
void ckref_gemm_f8_i8_mm(const uint8_t* a, const int8_t* b, const float* d0, const float* d1,
                         uint16_t* e, int M, int N, int K, int lda, int ldb, int lde, int fastgelu)
{
    auto ta = load2<f8_t>(a, M, K, lda, 1);
    auto tb = load2<int8_t>(b, K, N, 1, ldb);
    Tensor<float> tc(desc2(M, N, N, 1));
    using Ref = ck::tensor_operation::host::ReferenceGemm<f8_t, int8_t, float, float, PT, PT, PT>;
    auto ref  = Ref{};
    auto inv  = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(ta, tb, tc, PT{}, PT{}, PT{}));
    for(int m = 0; m < M; ++m)
        for(int n = 0; n < N; ++n)
        {
            half_t y;
            if(!fastgelu)
                ew::MultiplyMultiply{}(y, tc(m, n), d0[n], d1[m]);
            else
            {
                const float x = tc(m, n) * d0[n] * d1[m];
                ew::FastGelu{}(y, x);
            }
            e[static_cast<size_t>(m) * lde + n] = bits_to<uint16_t>(y);
        }
}

// example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:262-283, the host verification of the
// block-scaled problem, run through the reference's own pieces: type_convert<float>(f8_t), the fp32 dequantisation loops as
// the example writes them, ReferenceGemm<float, float, float, float>, type_convert to the output type.  B0 is int8 here
// (int4 / int8 weights); a1 row-major (ld_sa), b1 column-major (ld_sb) as the example's descriptors (:126-139).
void ckref_gemm_f8_i8_abscale(const uint8_t* a, const int8_t* b, const float* sa, const float* sb, uint16_t* e, int M, int N,
                              int K, int lda, int ldb, int lde, int bm, int bn, int bk, int ld_sa, int ld_sb, int e_bf16)
{
    auto ta = load2<f8_t>(a, M, K, lda, 1);
    auto tb = load2<int8_t>(b, K, N, 1, ldb);
    Tensor<float> a_m_k(desc2(M, K, K, 1)), b_k_n(desc2(K, N, N, 1)), c_m_n(desc2(M, N, N, 1));
    for(int m = 0; m < M; m++)
        for(int k = 0; k < K; k++)
            a_m_k(m, k) = ck::type_convert<float>(ta(m, k)) * sa[static_cast<size_t>(m / bm) * ld_sa + k / bk];
    for(int n = 0; n < N; n++)
        for(int k = 0; k < K; k++)
            b_k_n(k, n) = ck::type_convert<float>(tb(k, n)) * sb[static_cast<size_t>(k / bk) + static_cast<size_t>(n / bn) * ld_sb];
    using Ref = ck::tensor_operation::host::ReferenceGemm<float, float, float, float, PT, PT, PT>;
    auto ref  = Ref{};
    auto inv  = ref.MakeInvoker();
    inv.Run(ref.MakeArgument(a_m_k, b_k_n, c_m_n, PT{}, PT{}, PT{}));
    for(int m = 0; m < M; ++m)
        for(int n = 0; n < N; ++n)
        {
            if(e_bf16)
                e[static_cast<size_t>(m) * lde + n] = bits_to<uint16_t>(ck::type_convert<ck::bhalf_t>(c_m_n(m, n)));
            else
                e[static_cast<size_t>(m) * lde + n] = bits_to<uint16_t>(ck::type_convert<ck::half_t>(c_m_n(m, n)));
        }
}

} // extern "C"
