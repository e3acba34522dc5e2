// Host-side harness types for drivers written in the style of Composable Kernel's examples:
// strided host tensors, device buffers, the input generators, the tolerance check and the host
// reference GEMMs used by a driver's own "do_verification" step.  Fresh, compact restatements;
// the path-compatible headers under include/ck/library/ forward here.  None of this is on the GPU
// path: the device operators never call into it.
//
// Mirrors (paths relative to the CK tree):
//   HostTensorDescriptor / Tensor<T>      library/include/ck/library/utility/host_tensor.hpp:87-152,247-518
//   ParallelTensorFunctor (M x N form)    library/include/ck/library/utility/host_tensor.hpp:187-245
//   DeviceMem                             library/include/ck/library/utility/device_memory.hpp:20-39
//   FillUniformDistribution[IntegerValue] library/include/ck/library/utility/fill.hpp:18-86
//   GeneratorTensor_2 / _3                library/include/ck/library/utility/host_tensor_generator.hpp:84-169
//   check_err                             library/include/ck/library/utility/check_err.hpp:26-231
//   ReferenceGemm                         library/include/ck/library/reference_tensor_operation/cpu/reference_gemm.hpp:60-108
//   ReferencefpAintBGemm                  .../cpu/reference_fpAintB_gemm.hpp:62-127
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <iomanip>
#include <iostream>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

#include "wqgemm/ck_facade.hpp"

namespace ck {
namespace literals {
constexpr std::size_t operator""_uz(unsigned long long v) { return static_cast<std::size_t>(v); }
} // namespace literals
} // namespace ck

inline void hip_check_error(hipError_t e)
{
    if(e != hipSuccess)
        throw std::runtime_error(std::string("HIP runtime error: ") + hipGetErrorString(e));
}

struct HostTensorDescriptor
{
    HostTensorDescriptor() = default;
    HostTensorDescriptor(std::initializer_list<std::size_t> lens, std::initializer_list<std::size_t> strides)
        : mLens(lens), mStrides(strides)
    {
    }
    HostTensorDescriptor(std::initializer_list<std::size_t> lens) : mLens(lens)
    {
        mStrides.assign(mLens.size(), 1);
        for(int i = static_cast<int>(mLens.size()) - 2; i >= 0; --i)
            mStrides[i] = mStrides[i + 1] * mLens[i + 1];
    }
    std::size_t GetNumOfDimension() const { return mLens.size(); }
    std::size_t GetElementSize() const
    {
        return std::accumulate(mLens.begin(), mLens.end(), std::size_t{1}, std::multiplies<std::size_t>());
    }
    // 1 + sum (len - 1) * stride: a stride-0 dimension adds no storage
    std::size_t GetElementSpaceSize() const
    {
        std::size_t space = 1;
        for(std::size_t i = 0; i < mLens.size(); ++i)
            if(mLens[i] > 0)
                space += (mLens[i] - 1) * mStrides[i];
        return space;
    }
    const std::vector<std::size_t>& GetLengths() const { return mLens; }
    const std::vector<std::size_t>& GetStrides() const { return mStrides; }
    std::size_t Offset(std::size_t i0, std::size_t i1) const { return i0 * mStrides[0] + i1 * mStrides[1]; }

    friend std::ostream& operator<<(std::ostream& os, const HostTensorDescriptor& d)
    {
        os << "dim " << d.mLens.size() << ", lengths {";
        for(std::size_t i = 0; i < d.mLens.size(); ++i)
            os << (i ? ", " : "") << d.mLens[i];
        os << "}, strides {";
        for(std::size_t i = 0; i < d.mStrides.size(); ++i)
            os << (i ? ", " : "") << d.mStrides[i];
        return os << "}";
    }

    std::vector<std::size_t> mLens, mStrides;
};

// Run f(i0, i1) over a 2-D index space split into num_thread contiguous chunks of the flat range.
template <typename F>
void parallel_for_2d(F f, std::size_t n0, std::size_t n1, std::size_t num_thread)
{
    const std::size_t total = n0 * n1;
    num_thread              = std::max<std::size_t>(1, num_thread);
    const std::size_t per   = (total + num_thread - 1) / num_thread;
    std::vector<std::thread> threads;
    for(std::size_t t = 0; t < num_thread; ++t)
    {
        const std::size_t b = t * per, e = std::min(total, b + per);
        if(b >= e)
            break;
        auto body = [=] {
            for(std::size_t i = b; i < e; ++i)
                f(i / n1, i % n1);
        };
        if(num_thread == 1)
            body();
        else
            threads.emplace_back(body);
    }
    for(auto& th : threads)
        th.join();
}

template <typename T>
struct Tensor
{
    using Data = std::vector<T>;
    explicit Tensor(const HostTensorDescriptor& desc) : mDesc(desc), mData(desc.GetElementSpaceSize()) {}
    Tensor(std::initializer_list<std::size_t> lens) : mDesc(lens), mData(mDesc.GetElementSpaceSize()) {}

    T& operator()(std::size_t i0, std::size_t i1) { return mData[mDesc.Offset(i0, i1)]; }
    const T& operator()(std::size_t i0, std::size_t i1) const { return mData[mDesc.Offset(i0, i1)]; }
    // 1-D tensors (bias vectors)
    T& operator()(std::size_t i0) { return mData[i0 * mDesc.GetStrides()[0]]; }
    const T& operator()(std::size_t i0) const { return mData[i0 * mDesc.GetStrides()[0]]; }

    template <typename G>
    void GenerateTensorValue(G g, std::size_t num_thread = 1)
    {
        if(mDesc.GetNumOfDimension() == 1)
        {
            for(std::size_t i0 = 0; i0 < mDesc.GetLengths()[0]; ++i0)
                (*this)(i0) = g(i0);
            return;
        }
        parallel_for_2d([&](std::size_t i0, std::size_t i1) { (*this)(i0, i1) = g(i0, i1); },
                        mDesc.GetLengths()[0], mDesc.GetLengths()[1], num_thread);
    }
    template <typename U>
    Tensor<U> CopyAsType() const
    {
        Tensor<U> out(mDesc);
        std::transform(mData.begin(), mData.end(), out.mData.begin(),
                       [](const T& v) { return ck::type_convert<U>(v); });
        return out;
    }
    auto begin() { return mData.begin(); }
    auto end() { return mData.end(); }
    auto begin() const { return mData.begin(); }
    auto end() const { return mData.end(); }
    std::size_t GetElementSpaceSizeInBytes() const { return sizeof(T) * mData.size(); }

    HostTensorDescriptor mDesc;
    Data mData;
};

struct DeviceMem
{
    DeviceMem() = default;
    explicit DeviceMem(std::size_t bytes) : mMemSize(bytes) { hip_check_error(hipMalloc(&mpDeviceBuf, bytes)); }
    DeviceMem(const DeviceMem&) = delete;
    DeviceMem& operator=(const DeviceMem&) = delete;
    ~DeviceMem()
    {
        if(mpDeviceBuf)
            (void)hipFree(mpDeviceBuf);
    }
    void* GetDeviceBuffer() const { return mpDeviceBuf; }
    std::size_t GetBufferSize() const { return mMemSize; }
    void ToDevice(const void* p) const { hip_check_error(hipMemcpy(mpDeviceBuf, p, mMemSize, hipMemcpyHostToDevice)); }
    void FromDevice(void* p) const { hip_check_error(hipMemcpy(p, mpDeviceBuf, mMemSize, hipMemcpyDeviceToHost)); }
    void SetZero() const { hip_check_error(hipMemset(mpDeviceBuf, 0, mMemSize)); }

    void* mpDeviceBuf    = nullptr;
    std::size_t mMemSize = 0;
};

namespace ck {
namespace utils {

// A fresh mt19937(11939) per call; uniform_real_distribution<float>(a, b); optional std::round.
template <typename T>
struct FillUniformDistribution
{
    float a_{-5.f}, b_{5.f};
    template <typename Range>
    void operator()(Range&& r) const
    {
        std::mt19937 gen(11939);
        std::uniform_real_distribution<float> dis(a_, b_);
        for(auto& v : r)
            v = ck::type_convert<T>(dis(gen));
    }
};

template <typename T>
struct FillUniformDistributionIntegerValue
{
    float a_{-5.f}, b_{5.f};
    template <typename Range>
    void operator()(Range&& r) const
    {
        std::mt19937 gen(11939);
        std::uniform_real_distribution<float> dis(a_, b_);
        for(auto& v : r)
            v = ck::type_convert<T>(std::round(dis(gen)));
    }
};

// |out - ref| <= atol + rtol * |ref| element-wise; prints the first few offenders.
template <typename T>
bool check_err(const std::vector<T>& out, const std::vector<T>& ref,
               const std::string& msg = "Error: Incorrect results!", double rtol = -1, double atol = -1)
{
    if(rtol < 0)
        rtol = std::is_integral_v<T> ? 0 : (std::is_same_v<T, float> ? 1e-5 : 1e-3);
    if(atol < 0)
        atol = std::is_integral_v<T> ? 0 : (std::is_same_v<T, float> ? 3e-6 : 1e-3);
    if(out.size() != ref.size())
    {
        std::cerr << msg << " out.size() != ref.size()" << std::endl;
        return false;
    }
    std::size_t bad = 0;
    double max_err  = 0;
    for(std::size_t i = 0; i < out.size(); ++i)
    {
        const double o = ck::type_convert<float>(out[i]), r = ck::type_convert<float>(ref[i]);
        const double err = std::abs(o - r);
        if(err > atol + rtol * std::abs(r) || !std::isfinite(o) || !std::isfinite(r))
        {
            max_err = std::max(max_err, err);
            if(bad++ < 5)
                std::cerr << msg << std::setprecision(7) << " out[" << i << "] != ref[" << i << "]: " << o
                          << " != " << r << std::endl;
        }
    }
    if(bad)
        std::cerr << bad << " mismatches, max err: " << max_err << std::endl;
    return bad == 0;
}
template <typename T>
bool check_err(const Tensor<T>& out, const Tensor<T>& ref, const std::string& msg = "Error: Incorrect results!",
               double rtol = -1, double atol = -1)
{
    return check_err(out.mData, ref.mData, msg, rtol, atol);
}

} // namespace utils
} // namespace ck

template <typename T>
struct GeneratorTensor_2
{
    int min_value = 0, max_value = 1;
    template <typename... Is>
    T operator()(Is...)
    {
        const float v = static_cast<float>((std::rand() % (max_value - min_value)) + min_value);
        return ck::type_convert<T>(v);
    }
};
template <typename T>
struct GeneratorTensor_3
{
    float min_value = 0, max_value = 1;
    template <typename... Is>
    T operator()(Is...)
    {
        const float t = float(std::rand()) / float(RAND_MAX);
        return ck::type_convert<T>(min_value + t * (max_value - min_value));
    }
};

namespace ck {
namespace tensor_operation {
namespace host {

// c(m,n) = c_op( sum_k float(a_op(A(m,k))) * float(b_op(B(k,n))) ), k ascending, fp32 accumulate
template <typename ADataType, typename BDataType, typename CDataType, typename AccDataType,
          typename AElementwiseOperation, typename BElementwiseOperation, typename CElementwiseOperation>
struct ReferenceGemm
{
    struct Argument
    {
        const Tensor<ADataType>& a_m_k_;
        const Tensor<BDataType>& b_k_n_;
        Tensor<CDataType>& c_m_n_;
        AElementwiseOperation a_element_op_;
        BElementwiseOperation b_element_op_;
        CElementwiseOperation c_element_op_;
    };
    struct Invoker
    {
        float Run(const Argument& arg)
        {
            const std::size_t M = arg.c_m_n_.mDesc.GetLengths()[0], N = arg.c_m_n_.mDesc.GetLengths()[1];
            const std::size_t K = arg.a_m_k_.mDesc.GetLengths()[1];
            parallel_for_2d(
                [&](std::size_t m, std::size_t n) {
                    AccDataType acc = 0;
                    for(std::size_t k = 0; k < K; ++k)
                    {
                        CDataType va = 0, vb = 0; // compute types default to the C type
                        arg.a_element_op_(va, arg.a_m_k_(m, k));
                        arg.b_element_op_(vb, arg.b_k_n_(k, n));
                        acc += ck::type_convert<AccDataType>(va) * ck::type_convert<AccDataType>(vb);
                    }
                    CDataType vc = 0;
                    if constexpr(std::is_same_v<CDataType, int8_t> && !std::is_same_v<AccDataType, int32_t>)
                        arg.c_element_op_(vc, static_cast<int32_t>(acc)); // the int8 functor takes an int32
                    else
                        arg.c_element_op_(vc, acc);
                    arg.c_m_n_(m, n) = vc;
                },
                M, N, std::thread::hardware_concurrency());
            return 0;
        }
    };
    static Argument MakeArgument(const Tensor<ADataType>& a, const Tensor<BDataType>& b, Tensor<CDataType>& c,
                                 AElementwiseOperation a_op, BElementwiseOperation b_op, CElementwiseOperation c_op)
    {
        return Argument{a, b, c, a_op, b_op, c_op};
    }
    static Invoker MakeInvoker() { return Invoker{}; }
};

// c(m,n) = cvt_C( c_op( sum_k float(A(m,k)) * float( cvt_A(B(k,n)) * scale(k,n) ) ) )
template <typename ADataType, typename BDataType, typename ScaleDataType, typename CDataType, typename AccDataType,
          typename AElementwiseOperation, typename BElementwiseOperation, typename CElementwiseOperation>
struct ReferencefpAintBGemm
{
    struct Argument
    {
        const Tensor<ADataType>& a_m_k_;
        const Tensor<BDataType>& b_k_n_;
        const Tensor<ScaleDataType>& scale_k_n_;
        Tensor<CDataType>& c_m_n_;
        AElementwiseOperation a_element_op_;
        BElementwiseOperation b_element_op_;
        CElementwiseOperation c_element_op_;
    };
    struct Invoker
    {
        float Run(const Argument& arg)
        {
            const std::size_t M = arg.c_m_n_.mDesc.GetLengths()[0], N = arg.c_m_n_.mDesc.GetLengths()[1];
            const std::size_t K = arg.a_m_k_.mDesc.GetLengths()[1];
            parallel_for_2d(
                [&](std::size_t m, std::size_t n) {
                    AccDataType acc = 0;
                    for(std::size_t k = 0; k < K; ++k)
                    {
                        ADataType va;
                        BDataType vb;
                        ScaleDataType vs;
                        arg.a_element_op_(va, arg.a_m_k_(m, k));
                        arg.b_element_op_(vb, arg.b_k_n_(k, n));
                        arg.b_element_op_(vs, arg.scale_k_n_(k, n));
                        const ADataType deq = ck::type_convert<ADataType>(vb) * vs; // rounded to A's type
                        acc += ck::type_convert<AccDataType>(va) * ck::type_convert<AccDataType>(deq);
                    }
                    AccDataType vc;
                    arg.c_element_op_(vc, acc);
                    arg.c_m_n_(m, n) = ck::type_convert<CDataType>(vc);
                },
                M, N, std::thread::hardware_concurrency());
            return 0;
        }
    };
    static Argument MakeArgument(const Tensor<ADataType>& a, const Tensor<BDataType>& b,
                                 const Tensor<ScaleDataType>& s, Tensor<CDataType>& c, AElementwiseOperation a_op,
                                 BElementwiseOperation b_op, CElementwiseOperation c_op)
    {
        return Argument{a, b, s, c, a_op, b_op, c_op};
    }
    static Invoker MakeInvoker() { return Invoker{}; }
};

} // namespace host
} // namespace tensor_operation
} // namespace ck
