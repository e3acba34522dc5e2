// Client-API facade: a fresh, minimal restatement of the Composable Kernel types that cross the
// operator boundary of the weight-quantized GEMM path, so that callers written against CK's
// DeviceGemm* interfaces build against this tree alone.  The path-compatible headers under
// include/ck/ forward to this file.  When this engine is overlaid on a real CK checkout the
// concrete operators (device/impl/device_fpAintB_gemm_mfma_gfx950.hpp) include CK's own versions
// of these headers instead; signatures below are kept identical for that reason.
//
// What each block mirrors (paths relative to the CK tree):
//   index_t, half_t, bhalf_t, f8_t, int4_t      include/ck/ck.hpp:267-268, utility/data_type.hpp:10-14
//   type_convert                                include/ck/utility/type_convert.hpp:17-64
//   Tuple / Sequence / Number                   include/ck/utility/{tuple,sequence,number}.hpp (type lists only)
//   StreamConfig                                include/ck/stream_config.hpp:9-19
//   layout tags                                 tensor_operation/gpu/device/tensor_layout.hpp:13-24
//   GemmSpecialization                          tensor_operation/gpu/device/gemm_specialization.hpp:10-30
//   Base{Argument,Invoker,Operator}             tensor_operation/gpu/device/device_base.hpp:15-71
//   DeviceGemm, _dequantB, MultipleD, MultipleABD, SplitK, V2, V2R1
//                                               tensor_operation/gpu/device/device_gemm*.hpp
//   element-wise functors                       tensor_operation/gpu/element/*.hpp (host operator() only)
#pragma once

#include <hip/hip_runtime.h>

#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <typeinfo>

#include "wqgemm/wq_types.hpp"

namespace ck {

using index_t      = int32_t;
using long_index_t = int64_t;
using half_t       = _Float16;
using bhalf_t      = unsigned short;
using f8_t         = _BitInt(8); // E4M3 FNUZ bit pattern
using int4_t       = _BitInt(4); // one value per byte of storage, range -8..7

template <index_t N>
struct Number
{
    static constexpr index_t value = N;
    constexpr operator index_t() const { return N; }
};

template <index_t... Is>
struct Sequence
{
    static constexpr index_t Size() { return sizeof...(Is); }
};

template <typename... Ts>
struct Tuple
{
    static constexpr index_t Size() { return sizeof...(Ts); }
};

enum struct LoopScheduler
{
    Default,
    Interwave,
};
constexpr LoopScheduler make_default_loop_scheduler() { return LoopScheduler::Default; }

enum struct PipelineVersion
{
    v1,
    v2,
    v4,
    weight_only,
};

// ---- scalar conversions ---------------------------------------------------------------------
namespace detail {
inline float bits_to_float(uint32_t u)
{
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
inline uint32_t float_to_bits(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
} // namespace detail

// fp8 E4M3 FNUZ -> float: bias 8, 0x80 = NaN, no infinities, subnormals m * 2^-10
// (include/ck/utility/f8_utils.hpp:186-267, type_convert.hpp:383-396)
inline float f8_fnuz_to_float(uint8_t b)
{
    if(b == 0x80)
        return detail::bits_to_float(0x7F800001u);
    const int e   = (b >> 3) & 15, m = b & 7;
    const float v = e == 0 ? static_cast<float>(m) * 0.0009765625f : __builtin_ldexpf(static_cast<float>(8 + m), e - 11);
    return (b & 0x80) ? -v : v;
}

// fp8 E4M3FN (OCP, the gfx950 matrix core's E4M3; not a type of the reference): bias 7, S.1111.111 = NaN
inline float f8_e4m3fn_to_float(uint8_t b)
{
    if((b & 0x7F) == 0x7F)
        return detail::bits_to_float(0x7FC00000u);
    const int e   = (b >> 3) & 15, m = b & 7;
    const float v = e == 0 ? static_cast<float>(m) * 0.001953125f : __builtin_ldexpf(static_cast<float>(8 + m), e - 10);
    return (b & 0x80) ? -v : v;
}

template <typename Y, typename X>
inline Y type_convert(X x)
{
    if constexpr(std::is_same_v<X, f8_t>)
        return static_cast<Y>(f8_fnuz_to_float(static_cast<uint8_t>(x)));
    else if constexpr(std::is_same_v<X, wq_f8_e4m3fn_t>)
        return static_cast<Y>(f8_e4m3fn_to_float(x.data));
    else if constexpr(std::is_same_v<Y, bhalf_t> && std::is_same_v<X, float>)
        return static_cast<bhalf_t>(detail::float_to_bits(x) >> 16); // truncation
    else if constexpr(std::is_same_v<Y, float> && std::is_same_v<X, bhalf_t>)
        return detail::bits_to_float(static_cast<uint32_t>(x) << 16);
    else if constexpr(std::is_same_v<Y, bhalf_t> && (std::is_same_v<X, half_t> || std::is_same_v<X, int8_t>))
        return type_convert<bhalf_t>(static_cast<float>(x));
    else if constexpr(std::is_same_v<X, bhalf_t> && (std::is_same_v<Y, half_t> || std::is_same_v<Y, int8_t>))
        return static_cast<Y>(type_convert<float>(x));
    else
        return static_cast<Y>(x);
}

namespace tensor_layout {
struct BaseTensorLayout
{
};
namespace gemm {
struct RowMajor : BaseTensorLayout
{
    static constexpr const char* name = "RowMajor";
};
struct ColumnMajor : BaseTensorLayout
{
    static constexpr const char* name = "ColumnMajor";
};
} // namespace gemm
} // namespace tensor_layout

} // namespace ck

struct StreamConfig
{
    hipStream_t stream_id_ = nullptr;
    bool time_kernel_      = false;
    int log_level_         = 0;
    int cold_niters_       = 5;
    int nrepeat_           = 50;
    bool flush_cache       = false;
    int rotating_count     = 1;
};

namespace ck {
namespace tensor_operation {

// ---------------------------------------------------------------------------------------------
namespace element_wise {

struct PassThrough
{
    template <typename Y, typename X>
    void operator()(Y& y, const X& x) const
    {
        y = ck::type_convert<Y>(x);
    }
};

struct Relu
{
    template <typename T>
    void operator()(T& y, const T& x) const
    {
        y = x > T(0) ? x : T(0);
    }
};

// unary_element_wise_operation.hpp:236-251: float -> bhalf_t rounded to nearest even (type_convert.hpp:507-549) instead
// of type_convert's truncation; Inf / NaN patterns are not incremented, a NaN with its payload in the lower half keeps
// a mantissa bit.
struct ConvertBF16RTN
{
    template <typename Y, typename X>
    void operator()(Y& y, const X& x) const
    {
        static_assert(std::is_same_v<Y, bhalf_t>, "ConvertBF16RTN produces bhalf_t");
        const float f = ck::type_convert<float>(x);
        uint32_t u;
        std::memcpy(&u, &f, 4);
        if(~u & 0x7F800000u)
            u += 0x7FFFu + ((u >> 16) & 1u);
        else if(u & 0xFFFFu)
            u |= 0x10000u;
        y = static_cast<bhalf_t>(u >> 16);
    }
};

// y = x0 * x1 with the product formed in fp32 and narrowed by type_convert
struct Multiply
{
    template <typename Y, typename X0, typename X1>
    void operator()(Y& y, const X0& x0, const X1& x1) const
    {
        const float p = ck::type_convert<float>(x0) * ck::type_convert<float>(x1);
        y             = ck::type_convert<Y>(p);
    }
};

// y = x0 + x1 in fp32 (binary_element_wise_operation.hpp:12-115)
struct Add
{
    template <typename Y, typename X0, typename X1>
    void operator()(Y& y, const X0& x0, const X1& x1) const
    {
        y = ck::type_convert<Y>(ck::type_convert<float>(x0) + ck::type_convert<float>(x1));
    }
};

struct FastGelu
{
    // host form: u = x (c1 x^2 + c2), y = x / (1 + exp(u))
    template <typename Y, typename X>
    void operator()(Y& y, const X& x) const
    {
        const float xf  = ck::type_convert<float>(x);
        const float c1  = static_cast<float>(-2.0 * 0.035677f);
        const float c2  = static_cast<float>(-2.0 * 0.797885f);
        const float u   = xf * (c1 * xf * xf + c2);
        const float emu = static_cast<float>(std::exp(static_cast<double>(u)));
        y               = ck::type_convert<Y>(xf / (1.f + emu));
    }
};

// e = FastGelu(c + d) / FastGelu(c * d) in fp32 (binary_element_wise_operation.hpp:462-590)
struct AddFastGelu
{
    template <typename E, typename C, typename D>
    void operator()(E& e, const C& c, const D& d) const
    {
        float y = 0;
        FastGelu{}(y, ck::type_convert<float>(c) + ck::type_convert<float>(d));
        e = ck::type_convert<E>(y);
    }
};
struct MultiplyFastGelu
{
    template <typename E, typename C, typename D>
    void operator()(E& e, const C& c, const D& d) const
    {
        float y = 0;
        FastGelu{}(y, ck::type_convert<float>(c) * ck::type_convert<float>(d));
        e = ck::type_convert<E>(y);
    }
};

struct MultiplyMultiply
{
    template <typename E, typename C, typename D0, typename D1>
    void operator()(E& e, const C& c, const D0& d0, const D1& d1) const
    {
        const float x = ck::type_convert<float>(c) * ck::type_convert<float>(d0) * ck::type_convert<float>(d1);
        e             = ck::type_convert<E>(x);
    }
};

// element_wise_operation.hpp:200-231.  The reference specialises it per type tuple; for bhalf_t outputs and operands
// "type_convert<bhalf_t>(c) * d0 + d1" runs on unsigned short, i.e. on the bit patterns -- kept as written.
struct MultiplyAdd
{
    template <typename E, typename C, typename D0, typename D1>
    void operator()(E& e, const C& c, const D0& d0, const D1& d1) const
    {
        if constexpr(std::is_same_v<E, bhalf_t> && std::is_same_v<D0, bhalf_t> && std::is_same_v<D1, bhalf_t>)
            e = static_cast<bhalf_t>(static_cast<unsigned>(ck::type_convert<bhalf_t>(c)) * d0 + d1);
        else
            e = ck::type_convert<E>(ck::type_convert<float>(c) * ck::type_convert<float>(d0) + ck::type_convert<float>(d1));
    }
};

struct MultiplyAddFastGelu
{
    template <typename E, typename C, typename D0, typename D1>
    void operator()(E& e, const C& c, const D0& d0, const D1& d1) const
    {
        const float x0 = ck::type_convert<float>(c) * ck::type_convert<float>(d0) + ck::type_convert<float>(d1);
        float x1       = 0;
        FastGelu{}(x1, x0);
        e = ck::type_convert<E>(x1);
    }
};

// Not in CK: the fused epilogue of the fp8 x int4 path, e = FastGelu(c * d0 * d1).
struct MultiplyMultiplyFastGelu
{
    template <typename E, typename C, typename D0, typename D1>
    void operator()(E& e, const C& c, const D0& d0, const D1& d1) const
    {
        const float x0 = ck::type_convert<float>(c) * ck::type_convert<float>(d0) * ck::type_convert<float>(d1);
        float x1       = 0;
        FastGelu{}(x1, x0);
        e = ck::type_convert<E>(x1);
    }
};

template <typename Activation>
struct Activation_Mul_Clamp
{
    Activation_Mul_Clamp(float requantScale, Activation activationOp)
        : requantScale_(requantScale), activationOp_(activationOp)
    {
    }
    void operator()(int8_t& y, const int32_t& x) const
    {
        float v = static_cast<float>(x);
        activationOp_(v, v);
        v = std::fmin(std::fmax(requantScale_ * v, -128.f), 127.f);
        y = static_cast<int8_t>(v);
    }
    float requantScale_;
    Activation activationOp_;
};

template <typename Activation>
struct Add_Activation_Mul_Clamp
{
    Add_Activation_Mul_Clamp(float requantScale, Activation activationOp)
        : requantScale_(requantScale), activationOp_(activationOp)
    {
    }
    void operator()(int8_t& y, const int32_t& x, const int32_t& bias) const
    {
        float v = static_cast<float>(x + bias);
        activationOp_(v, v);
        v = std::fmin(std::fmax(requantScale_ * v, -128.f), 127.f);
        y = static_cast<int8_t>(v);
    }
    float requantScale_;
    Activation activationOp_;
};

} // namespace element_wise

// ---------------------------------------------------------------------------------------------
namespace device {

enum struct GemmSpecialization
{
    Default,
    MPadding,
    NPadding,
    KPadding,
    MNPadding,
    MKPadding,
    NKPadding,
    MNKPadding,
    OPadding,
    MOPadding,
    NOPadding,
    KOPadding,
    MNOPadding,
    MKOPadding,
    NKOPadding,
    MNKOPadding,
};

inline std::string getGemmSpecializationString(const GemmSpecialization& s)
{
    static const char* const names[] = {"Default",   "MPadding",   "NPadding",   "KPadding",
                                        "MNPadding", "MKPadding",  "NKPadding",  "MNKPadding",
                                        "OPadding",  "MOPadding",  "NOPadding",  "KOPadding",
                                        "MNOPadding", "MKOPadding", "NKOPadding", "MNKOPadding"};
    const int i                     = static_cast<int>(s);
    return i >= 0 && i < 16 ? names[i] : "Unrecognized specialization!";
}

struct BaseArgument
{
    BaseArgument()                    = default;
    BaseArgument(const BaseArgument&) = default;
    BaseArgument& operator=(const BaseArgument&) = default;
    virtual ~BaseArgument() {}

    void* p_workspace_ = nullptr;
};

struct BaseInvoker
{
    BaseInvoker()                   = default;
    BaseInvoker(const BaseInvoker&) = default;
    BaseInvoker& operator=(const BaseInvoker&) = default;
    virtual ~BaseInvoker() {}

    virtual float Run(const BaseArgument*, const StreamConfig& = StreamConfig{}) { return 0.f; }
};

struct BaseOperator
{
    BaseOperator()                    = default;
    BaseOperator(const BaseOperator&) = default;
    BaseOperator& operator=(const BaseOperator&) = default;
    virtual ~BaseOperator() {}

    virtual bool IsSupportedArgument(const BaseArgument*) { return false; }
    virtual std::string GetTypeString() const { return ""; }
    virtual std::string GetTypeIdName() const { return typeid(*this).name(); }
    virtual std::string GetTypeIdHashCode() const
    {
        std::ostringstream os;
        os << std::hex << typeid(*this).hash_code();
        return os.str();
    }
    virtual size_t GetWorkSpaceSize(const BaseArgument*) const { return 0; }
    virtual void SetWorkSpacePointer(BaseArgument* p_arg, void* p_workspace,
                                     const StreamConfig& = StreamConfig{}) const
    {
        assert(p_arg);
        p_arg->p_workspace_ = p_workspace;
    }
};

template <typename ALayout, typename BLayout, typename CLayout, typename ADataType, typename BDataType,
          typename CDataType, typename AElementwiseOperation, typename BElementwiseOperation,
          typename CElementwiseOperation>
struct DeviceGemm : public BaseOperator
{
    virtual std::unique_ptr<BaseArgument>
    MakeArgumentPointer(const void* p_a, const void* p_b, void* p_c, ck::index_t M, ck::index_t N,
                        ck::index_t K, ck::index_t StrideA, ck::index_t StrideB, ck::index_t StrideC,
                        AElementwiseOperation a_element_op, BElementwiseOperation b_element_op,
                        CElementwiseOperation c_element_op) = 0;
    virtual std::unique_ptr<BaseInvoker> MakeInvokerPointer() = 0;
};

template <typename ALayout, typename BLayout, typename CLayout, typename ADataType, typename BDataType,
          typename CDataType, typename AElementwiseOperation, typename BElementwiseOperation,
          typename CElementwiseOperation>
struct DeviceGemm_dequantB : public BaseOperator
{
    virtual std::unique_ptr<BaseArgument>
    MakeArgumentPointer(const void* p_a, const void* p_b, const void* p_scale, void* p_c, ck::index_t M,
                        ck::index_t N, ck::index_t K, ck::index_t StrideA, ck::index_t StrideB,
                        ck::index_t StrideC, AElementwiseOperation a_element_op,
                        BElementwiseOperation b_element_op, CElementwiseOperation c_element_op) = 0;
    virtual std::unique_ptr<BaseInvoker> MakeInvokerPointer() = 0;
};

template <typename ALayout, typename BLayout, typename DsLayout, typename ELayout, typename ADataType,
          typename BDataType, typename DsDataType, typename EDataType, typename AElementwiseOperation,
          typename BElementwiseOperation, typename CDEElementwiseOperation>
struct DeviceGemmMultipleD : public BaseOperator
{
    static constexpr index_t NumDTensor = DsDataType::Size();
    virtual std::unique_ptr<BaseArgument>
    MakeArgumentPointer(const void* p_a, const void* p_b, std::array<const void*, NumDTensor> p_ds, void* p_e,
                        ck::index_t M, ck::index_t N, ck::index_t K, ck::index_t StrideA, ck::index_t StrideB,
                        std::array<ck::index_t, NumDTensor> StrideDs, ck::index_t StrideE,
                        AElementwiseOperation a_element_op, BElementwiseOperation b_element_op,
                        CDEElementwiseOperation cde_element_op) = 0;
    virtual std::unique_ptr<BaseInvoker> MakeInvokerPointer() = 0;
};

// include/ck/tensor_operation/gpu/device/device_gemm_multiple_d_ab_scale.hpp:23-63
template <typename ALayout, typename BLayout, typename DsLayout, typename ELayout, typename ADataType,
          typename AScaleType, typename BDataType, typename BScaleType, typename DsDataType, typename EDataType,
          index_t ScaleBlockM, index_t ScaleBlockN, index_t ScaleBlockK, typename AElementwiseOperation,
          typename BElementwiseOperation, typename CDEElementwiseOperation>
struct DeviceGemmMultipleD_ABScale : public BaseOperator
{
    static constexpr index_t NumDTensor = DsDataType::Size();
    virtual std::unique_ptr<BaseArgument>
    MakeArgumentPointer(const void* p_a, const void* p_b, std::array<const void*, NumDTensor> p_ds, void* p_e,
                        const ck::index_t M, const ck::index_t N, const ck::index_t K, const ck::index_t StrideA,
                        const ck::index_t StrideB, const std::array<ck::index_t, NumDTensor> StrideDs,
                        const ck::index_t StrideE, const void* p_a_scale, const void* p_b_scale,
                        AElementwiseOperation a_element_op, BElementwiseOperation b_element_op,
                        CDEElementwiseOperation cde_element_op) = 0;
    virtual std::unique_ptr<BaseInvoker> MakeInvokerPointer() = 0;
};

template <typename AsLayout, typename BsLayout, typename DsLayout, typename ELayout, typename AsDataType,
          typename BsDataType, typename DsDataType, typename EDataType, typename AElementwiseOperation,
          typename BElementwiseOperation, typename CDEElementwiseOperation>
struct DeviceGemmMultipleABD : public BaseOperator
{
    static constexpr index_t NumATensor = AsDataType::Size();
    static constexpr index_t NumBTensor = BsDataType::Size();
    static constexpr index_t NumDTensor = DsDataType::Size();
    virtual std::unique_ptr<BaseArgument>
    MakeArgumentPointer(std::array<const void*, NumATensor> p_as, std::array<const void*, NumBTensor> p_bs,
                        std::array<const void*, NumDTensor> p_ds, void* p_e, ck::index_t M, ck::index_t N,
                        ck::index_t K, std::array<ck::index_t, NumATensor> StrideAs,
                        std::array<ck::index_t, NumBTensor> StrideBs,
                        std::array<ck::index_t, NumDTensor> StrideDs, ck::index_t StrideE,
                        AElementwiseOperation a_element_op, BElementwiseOperation b_element_op,
                        CDEElementwiseOperation cde_element_op) = 0;
    virtual std::unique_ptr<BaseInvoker> MakeInvokerPointer() = 0;
};

template <typename ALayout, typename BLayout, typename CLayout, typename ADataType, typename BDataType,
          typename CDataType, typename AElementwiseOperation, typename BElementwiseOperation,
          typename CElementwiseOperation, typename ComputeType = CDataType>
struct DeviceGemmSplitK : public BaseOperator
{
    virtual std::unique_ptr<BaseArgument>
    MakeArgumentPointer(const void* p_a, const void* p_b, void* p_c, ck::index_t M, ck::index_t N,
                        ck::index_t K, ck::index_t StrideA, ck::index_t StrideB, ck::index_t StrideC,
                        AElementwiseOperation a_element_op, BElementwiseOperation b_element_op,
                        CElementwiseOperation c_element_op, ck::index_t KBatch) = 0;
    virtual std::unique_ptr<BaseInvoker> MakeInvokerPointer() = 0;
};

template <typename ALayout, typename BLayout, typename CLayout, typename ADataType, typename BDataType,
          typename CDataType, typename AElementwiseOperation, typename BElementwiseOperation,
          typename CElementwiseOperation>
struct DeviceGemmV2 : public BaseOperator
{
    virtual std::unique_ptr<BaseArgument>
    MakeArgumentPointer(const void* p_a, const void* p_b, void* p_c, ck::index_t M, ck::index_t N,
                        ck::index_t K, ck::index_t StrideA, ck::index_t StrideB, ck::index_t StrideC,
                        ck::index_t KSplit, AElementwiseOperation a_element_op,
                        BElementwiseOperation b_element_op, CElementwiseOperation c_element_op) = 0;
    virtual std::unique_ptr<BaseInvoker> MakeInvokerPointer() = 0;
};

template <typename ALayout, typename BLayout, typename DsLayout, typename CLayout, typename ADataType,
          typename BDataType, typename DsDataType, typename CDataType, typename AElementwiseOperation,
          typename BElementwiseOperation, typename CElementwiseOperation>
struct DeviceGemmV2R1 : public BaseOperator
{
    static constexpr index_t NumDTensor = DsDataType::Size();
    virtual std::unique_ptr<BaseArgument>
    MakeArgumentPointer(const void* p_a, const void* p_b, std::array<const void*, NumDTensor> p_ds, void* p_c,
                        ck::index_t M, ck::index_t N, ck::index_t K, ck::index_t StrideA, ck::index_t StrideB,
                        std::array<ck::index_t, NumDTensor> DsStrides, ck::index_t StrideC, ck::index_t KSplit,
                        AElementwiseOperation a_element_op, BElementwiseOperation b_element_op,
                        CElementwiseOperation c_element_op) = 0;
    virtual std::unique_ptr<BaseInvoker> MakeInvokerPointer() = 0;
};

} // namespace device
} // namespace tensor_operation
} // namespace ck
