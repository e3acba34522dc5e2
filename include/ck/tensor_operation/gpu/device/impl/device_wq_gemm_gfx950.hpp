// Concrete gfx950 (MI355X) operators of the weight-quantized GEMM engine behind Composable
// Kernel's abstract client interfaces.  Host-only C++: every operator captures the problem into a
// `wqgemm_problem` and calls the C ABI of libwqgemm.so (include/wqgemm.h); no device code is
// compiled from this header, so a client needs only a host compiler and the HIP runtime headers.
//
// The header includes the abstract interfaces by their CK paths.  Built against this tree they
// resolve to the facade (include/wqgemm/ck_facade.hpp); overlaid on a CK checkout they resolve to
// CK's own headers, and these operators become additional instances next to CK's.
//
//   DeviceFpAintBGemm_Mfma_gfx950        : DeviceGemm_dequantB     (example/64_fpAintB_gemm)
//        stands where DeviceFpAintBGemm_Wmma_CShuffle does
//        (include/ck/tensor_operation/gpu/device/impl/device_fpAintB_gemm_wmma.hpp:70-710)
//   DeviceGemmQuantization_gfx950        : DeviceGemmMultipleD     (example/14, client_example/09)
//        stands where DeviceGemmMultipleD_Xdl_CShuffle<int8...> does
//        (impl/device_gemm_multiple_d_xdl_cshuffle.hpp)
//   DeviceGemmMultiABD_WeightQuant_gfx950: DeviceGemmMultipleABD   (example/60, client_example/30)
//        stands where DeviceGemmMultipleABD_Xdl_CShuffle<(bf16),(int8,bf16)> does
//        (impl/device_gemm_multiple_abd_xdl_cshuffle.hpp)
//   DeviceGemmMultiplyMultiply_WeightQuant_gfx950 : DeviceGemmMultipleD (example/65)
//        stands where DeviceGemmMultiD_Xdl_CShuffle_V3<f8,f8,(f32,f32)> does
//        (impl/device_gemm_multiple_d_xdl_cshuffle_v3.hpp)
//
//   DeviceGemmMultiD_ABScale_WeightQuant_gfx950 : DeviceGemmMultipleD_ABScale (example/65 ..._fp8_ab_scale)
//        stands where DeviceGemmMultiD_ABScale_Xdl_CShuffle_V3<f8, f32, f8, f32> does
//        (impl/device_gemm_multiple_d_xdl_cshuffle_v3_ab_scale.hpp), with int4 / int8 weights in place of fp8 ones
//
//   DeviceGemmV2R1_WeightQuant_gfx950    : DeviceGemmV2R1          (example/35 gemm_xdl_splitk_reduce_bf16A_i8B)
//        stands where DeviceGemm_Xdl_CShuffleV3R1<bf16, int8> does
//        (impl/device_gemm_xdl_cshuffle_v3r1.hpp:28-700): KSplit before the functors, fp32 slabs in the
//        caller's workspace (GetWorkSpaceSize / SetWorkSpacePointer), reduce + convert afterwards
//   DeviceGemmSplitK_WeightQuant_gfx950  : DeviceGemmSplitK        (profiler gemm_splitk: trailing KBatch)
//
// Conventions kept from the reference (device_fpAintB_gemm_wmma.hpp:424-433,497-501,594-630):
// Argument holds raw non-owning pointers and copies of the functors; IsSupportedArgument returns
// false for problems the engine cannot run; Invoker::Run throws std::runtime_error on an
// unsupported argument and on HIP failures, returns the mean kernel time in ms when
// StreamConfig::time_kernel_ is set and 0 after one asynchronous launch otherwise.
#pragma once

#include <hip/hip_runtime.h>

#include <array>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "ck/ck.hpp"
#include "ck/utility/data_type.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_dequantB.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_multiple_d.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_multiple_d_ab_scale.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_multiple_abd.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_splitk.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_v2.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"

#include "wqgemm.h"
#include "wqgemm/wq_types.hpp"

namespace ck {

// Storage tag for packed int4 weights: two offset-binary nibbles per byte in the layout
// documented at wqgemm_pack_int4 (include/wqgemm.h).  Strides of such a tensor still count
// logical elements.
struct wq_int4x2_t
{
    uint8_t data;
};
// The same nibbles, and plain int8 weights, in their pre-shuffled order (wqgemm_preshuffle_int4 / _int8 in
// include/wqgemm.h): written once offline; StrideB is the K pitch handed to the re-layout.
struct wq_int4x2_preshuffled_t
{
    uint8_t data;
};
struct wq_int8_preshuffled_t
{
    int8_t data;
};

namespace tensor_operation {
namespace device {

namespace wq_detail {

template <typename T>
struct dtype_of;
template <>
struct dtype_of<ck::half_t>
{
    static constexpr int value = WQ_F16;
};
template <>
struct dtype_of<ck::bhalf_t>
{
    static constexpr int value = WQ_BF16;
};
template <>
struct dtype_of<ck::f8_t>
{
    static constexpr int value = WQ_F8_FNUZ;
};
template <>
struct dtype_of<ck::wq_f8_e4m3fn_t>
{
    static constexpr int value = WQ_F8_E4M3FN;
};
template <>
struct dtype_of<float>
{
    static constexpr int value = WQ_F32;
};
template <>
struct dtype_of<int8_t>
{
    static constexpr int value = WQ_I8;
};
template <>
struct dtype_of<uint8_t>
{
    static constexpr int value = WQ_U8_OFFSET;
};
template <>
struct dtype_of<ck::wq_int4x2_t>
{
    static constexpr int value = WQ_I4_PACKED;
};
template <>
struct dtype_of<ck::wq_int4x2_preshuffled_t>
{
    static constexpr int value = WQ_I4_PRESHUFFLED;
};
template <>
struct dtype_of<ck::wq_int8_preshuffled_t>
{
    static constexpr int value = WQ_I8_PRESHUFFLED;
};
template <>
struct dtype_of<int32_t>
{
    static constexpr int value = WQ_I32;
};

template <typename Layout>
constexpr int layout_of()
{
    return std::is_same_v<Layout, ck::tensor_layout::gemm::RowMajor> ? WQ_ROW_MAJOR : WQ_COL_MAJOR;
}

template <typename Op>
constexpr int unary_epilogue_of()
{
    namespace ew = ck::tensor_operation::element_wise;
    if constexpr(std::is_same_v<Op, ew::PassThrough>)
        return WQ_EPI_PASSTHROUGH;
    else if constexpr(std::is_same_v<Op, ew::FastGelu>)
        return WQ_EPI_FASTGELU;
    else
        return -1;
}

inline bool device_is_gfx950()
{
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess)
        return false;
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

inline bool supported(const wqgemm_problem& prob)
{
    // a client compiled against another release of include/wqgemm.h than the library it found at run time
    if(wqgemm_problem_size() != sizeof(wqgemm_problem))
        throw std::runtime_error("libwqgemm.so and include/wqgemm.h disagree on sizeof(wqgemm_problem): rebuild the client");
    if(!device_is_gfx950())
    {
        std::cout << "DeviceOp err: Arch" << std::endl; // the reference's wording for a wrong target
        return false;
    }
    if(wqgemm_is_supported(&prob) != 0)
        return true;
    std::cerr << "wqgemm: " << wqgemm_last_error() << std::endl; // why, for the caller's "does not support" line
    return false;
}

inline float run(const wqgemm_problem& prob, void* workspace, const StreamConfig& cfg, const char* who)
{
    if(wqgemm_is_supported(&prob) == 0)
        throw std::runtime_error(std::string("wrong! ") + who + " has invalid setting: " + wqgemm_last_error());
    if(cfg.time_kernel_)
    {
        float ms = 0.f;
        // StreamConfig::flush_cache selects the reference's other timing mode: rotating copies of A and B, one event
        // pair per launch, median (include/ck/host_utility/flush_cache.hpp:107-235)
        const int rc = cfg.flush_cache
                           ? wqgemm_run_timed_rotating(&prob, workspace, cfg.stream_id_, cfg.cold_niters_, cfg.nrepeat_,
                                                       cfg.rotating_count, &ms)
                           : wqgemm_run_timed(&prob, workspace, cfg.stream_id_, cfg.cold_niters_, cfg.nrepeat_, &ms);
        if(rc != WQ_OK)
            throw std::runtime_error(std::string(who) + ": " + wqgemm_last_error());
        return ms;
    }
    const int rc = wqgemm_launch(&prob, workspace, cfg.stream_id_);
    if(rc != WQ_OK)
        throw std::runtime_error(std::string(who) + ": " + wqgemm_last_error());
    return 0.f;
}

inline std::string type_string(const wqgemm_problem& prob)
{
    char buf[512];
    wqgemm_type_string(&prob, buf, sizeof(buf));
    return buf;
}

// Device memory for the arguments' own scratch, recycled: an Argument takes a block when it is made and hands it back when
// its last copy goes, so a caller that makes arguments in a loop (the profilers do, once per instance and problem) meets
// hipMalloc only until the pool holds a block that fits.  Every Argument still has a block of its own while it lives
// (arguments used at the same time on different streams never share scratch).  A block that comes back may still be read
// or written by launches in flight on any stream -- the pool does not know which -- so a block is handed out again only
// behind a hipDeviceSynchronize of its device (microseconds on an idle device, and no more of a synchronisation than the
// hipMalloc it replaces).  Blocks are kept per device, at most kMaxIdle of them and kMaxIdleBytes in all; what exceeds that
// is freed at once.  The pool itself is never destroyed (at process exit the HIP runtime may be gone before static
// destructors run).
class ScratchPool
{
    public:
    static ScratchPool& instance()
    {
        static ScratchPool* pool = new ScratchPool;
        return *pool;
    }
    std::shared_ptr<void> take(size_t bytes)
    {
        int dev = 0;
        if(hipGetDevice(&dev) != hipSuccess)
            return nullptr;
        Block blk{nullptr, 0, dev};
        {
            std::lock_guard<std::mutex> lock(mu_);
            // the smallest idle block of this device that fits and is not more than twice as large
            int best = -1;
            for(int i = 0; i < (int)idle_.size(); ++i)
                if(idle_[i].dev == dev && idle_[i].bytes >= bytes && idle_[i].bytes <= 2 * bytes + (1u << 20) &&
                   (best < 0 || idle_[i].bytes < idle_[best].bytes))
                    best = i;
            if(best >= 0)
            {
                blk = idle_[best];
                idle_bytes_ -= blk.bytes;
                idle_.erase(idle_.begin() + best);
            }
        }
        if(blk.ptr)
        {
            // whatever used the block before it came back has to be done before its next owner writes to it
            if(hipDeviceSynchronize() != hipSuccess)
                (void)hipGetLastError();
        }
        else
        {
            if(hipMalloc(&blk.ptr, bytes) != hipSuccess)
            {
                (void)hipGetLastError();
                return nullptr;
            }
            blk.bytes = bytes;
        }
        return std::shared_ptr<void>(blk.ptr, [blk](void*) { ScratchPool::instance().give(blk); });
    }

    private:
    struct Block
    {
        void* ptr;
        size_t bytes;
        int dev;
    };
    static constexpr size_t kMaxIdle = 8, kMaxIdleBytes = size_t(2) << 30;
    void give(Block blk)
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            if(idle_.size() < kMaxIdle && idle_bytes_ + blk.bytes <= kMaxIdleBytes)
            {
                idle_.push_back(blk);
                idle_bytes_ += blk.bytes;
                return;
            }
        }
        // (the block may still be in use by launches in flight: hipFree waits for the device)
        int cur = 0;
        const bool swap = hipGetDevice(&cur) == hipSuccess && cur != blk.dev && hipSetDevice(blk.dev) == hipSuccess;
        (void)hipFree(blk.ptr);
        if(swap)
            (void)hipSetDevice(cur);
    }
    std::mutex mu_;
    std::vector<Block> idle_;
    size_t idle_bytes_ = 0;
};

// Argument / Invoker shared by every operator below.
struct Argument : public BaseArgument
{
    wqgemm_problem prob{};
    // Scratch for the optional operand re-layouts of the engine (fp16 image of FNUZ activations, K-contiguous copy
    // of row-major 8-bit weights), owned by the argument so that clients which never heard of a workspace
    // (client_example/30_gemm_bf16Aint8B) still reach the matrix-core kernels.  A workspace handed over with
    // SetWorkSpacePointer takes precedence; split-K workspaces stay the caller's (device_base.hpp:60-67).
    std::shared_ptr<void> owned_workspace;
    // with_splitk_slabs: also for k_batch > 1, for interfaces whose clients have no workspace call at all
    // (DeviceGemmSplitK: profiler/include/profiler/profile_gemm_splitk_impl.hpp:160-176)
    void AllocateOptionalWorkspace(bool with_splitk_slabs = false)
    {
        if(prob.k_batch > 1 && !with_splitk_slabs)
            return;
        const size_t bytes = wqgemm_workspace_bytes(&prob);
        if(bytes == 0)
            return;
        owned_workspace = ScratchPool::instance().take(bytes);
        if(!owned_workspace)
        {
            static bool told = false;
            if(!told)
                std::cerr << "wqgemm: could not allocate " << bytes << " bytes of operator-owned scratch: "
                          << (prob.k_batch > 1 ? "split-K cannot run" : "the general kernel will run, far slower")
                          << "; hand a workspace over with SetWorkSpacePointer" << std::endl;
            told = true;
        }
    }
    void* Workspace() const { return p_workspace_ ? p_workspace_ : owned_workspace.get(); }
};

struct Invoker : public BaseInvoker
{
    explicit Invoker(const char* who) : who_(who) {}
    float Run(const Argument& arg, const StreamConfig& cfg = StreamConfig{})
    {
        return run(arg.prob, arg.Workspace(), cfg, who_);
    }
    float Run(const BaseArgument* p_arg, const StreamConfig& cfg = StreamConfig{}) override
    {
        return Run(*dynamic_cast<const Argument*>(p_arg), cfg);
    }
    const char* who_;
};

} // namespace wq_detail

// ---------------------------------------------------------------------------------------------
// C[M,N] = c_op( A[M,K] * (cvt_A(B[K,N]) * scale[N]) )      fp16 / bf16 activations
// ---------------------------------------------------------------------------------------------
template <typename ALayout, typename BLayout, typename CLayout, typename ADataType, typename BDataType,
          typename ScaleDataType, typename CDataType, typename AElementwiseOperation,
          typename BElementwiseOperation, typename CElementwiseOperation>
struct DeviceFpAintBGemm_Mfma_gfx950
    : public DeviceGemm_dequantB<ALayout, BLayout, CLayout, ADataType, BDataType, CDataType,
                                 AElementwiseOperation, BElementwiseOperation, CElementwiseOperation>
{
    using Argument = wq_detail::Argument;
    using Invoker  = wq_detail::Invoker;

    static_assert(wq_detail::unary_epilogue_of<CElementwiseOperation>() >= 0,
                  "CElementwiseOperation must be PassThrough or FastGelu");

    static Argument MakeArgument(const ADataType* p_a, const BDataType* p_b, const ScaleDataType* p_scale,
                                 CDataType* p_c, index_t M, index_t N, index_t K, index_t StrideA,
                                 index_t StrideB, index_t StrideC, AElementwiseOperation,
                                 BElementwiseOperation, CElementwiseOperation, index_t KBatch = 1)
    {
        Argument arg;
        wqgemm_problem& p = arg.prob;
        p.a_dtype         = wq_detail::dtype_of<ADataType>::value;
        p.b_dtype         = wq_detail::dtype_of<BDataType>::value;
        p.scale_dtype     = wq_detail::dtype_of<ScaleDataType>::value;
        p.c_dtype         = wq_detail::dtype_of<CDataType>::value;
        p.a_layout        = wq_detail::layout_of<ALayout>();
        p.b_layout        = wq_detail::layout_of<BLayout>();
        p.c_layout        = wq_detail::layout_of<CLayout>();
        p.M = M, p.N = N, p.K = K;
        p.lda = StrideA, p.ldb = StrideB, p.ldc = StrideC;
        p.a = p_a, p.b = p_b, p.scale = p_scale, p.c = p_c;
        p.epilogue = wq_detail::unary_epilogue_of<CElementwiseOperation>();
        p.k_batch  = KBatch;
        return arg;
    }

    static Invoker MakeInvoker() { return Invoker{"DeviceFpAintBGemm_Mfma_gfx950"}; }

    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }

    std::unique_ptr<BaseArgument> MakeArgumentPointer(const void* p_a, const void* p_b, const void* p_scale,
                                                      void* p_c, index_t M, index_t N, index_t K,
                                                      index_t StrideA, index_t StrideB, index_t StrideC,
                                                      AElementwiseOperation a_op, BElementwiseOperation b_op,
                                                      CElementwiseOperation c_op) override
    {
        return std::make_unique<Argument>(MakeArgument(
            static_cast<const ADataType*>(p_a), static_cast<const BDataType*>(p_b),
            static_cast<const ScaleDataType*>(p_scale), static_cast<CDataType*>(p_c), M, N, K, StrideA,
            StrideB, StrideC, a_op, b_op, c_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }

    size_t GetWorkSpaceSize(const BaseArgument* p_arg) const override
    {
        return wqgemm_workspace_bytes(&dynamic_cast<const Argument*>(p_arg)->prob);
    }

    std::string GetTypeString() const override
    {
        wqgemm_problem p{};
        p.a_dtype = wq_detail::dtype_of<ADataType>::value;
        p.b_dtype = wq_detail::dtype_of<BDataType>::value;
        p.c_dtype = wq_detail::dtype_of<CDataType>::value;
        return std::string("DeviceFpAintBGemm_Mfma_gfx950<") + (p.a_dtype == WQ_BF16 ? "BF16" : "F16") + ", " +
               (p.b_dtype == WQ_I4_PACKED        ? "I4x2"
                : p.b_dtype == WQ_I4_PRESHUFFLED ? "I4_PRESHUFFLED"
                : p.b_dtype == WQ_I8_PRESHUFFLED ? "I8_PRESHUFFLED"
                : p.b_dtype == WQ_I8             ? "I8"
                                                 : "U8") +
               ", LdsDma, DequantInRegister, Mfma16x16x32>";
    }
};

// ---------------------------------------------------------------------------------------------
// E[M,N] = Activation_Mul_Clamp( A[M,K] int8 * B[K,N] int8 )              int8 requantised GEMM
// ---------------------------------------------------------------------------------------------
template <typename ALayout, typename BLayout, typename ELayout, typename CDEElementwiseOperation>
struct DeviceGemmQuantization_gfx950
    : public DeviceGemmMultipleD<ALayout, BLayout, ck::Tuple<>, ELayout, int8_t, int8_t, ck::Tuple<>, int8_t,
                                 element_wise::PassThrough, element_wise::PassThrough, CDEElementwiseOperation>
{
    using Argument    = wq_detail::Argument;
    using Invoker     = wq_detail::Invoker;
    using PassThrough = element_wise::PassThrough;

    static Argument MakeArgument(const void* p_a, const void* p_b, std::array<const void*, 0>, void* p_e,
                                 index_t M, index_t N, index_t K, index_t StrideA, index_t StrideB,
                                 std::array<index_t, 0>, index_t StrideE, PassThrough, PassThrough,
                                 CDEElementwiseOperation cde_op)
    {
        Argument arg;
        wqgemm_problem& p = arg.prob;
        p.a_dtype = p.b_dtype = p.c_dtype = WQ_I8;
        p.scale_dtype                     = WQ_F32;
        p.a_layout                        = wq_detail::layout_of<ALayout>();
        p.b_layout                        = wq_detail::layout_of<BLayout>();
        p.c_layout                        = wq_detail::layout_of<ELayout>();
        p.M = M, p.N = N, p.K = K;
        p.lda = StrideA, p.ldb = StrideB, p.ldc = StrideE;
        p.a = p_a, p.b = p_b, p.c = p_e;
        p.epilogue      = std::is_same_v<CDEElementwiseOperation, element_wise::Activation_Mul_Clamp<element_wise::Relu>>
                              ? WQ_EPI_RELU_MUL_CLAMP
                              : WQ_EPI_ACT_MUL_CLAMP;
        p.requant_scale = cde_op.requantScale_;
        p.k_batch       = 1;
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{"DeviceGemmQuantization_gfx950"}; }
    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument> MakeArgumentPointer(const void* p_a, const void* p_b,
                                                      std::array<const void*, 0> p_ds, void* p_e, index_t M,
                                                      index_t N, index_t K, index_t StrideA, index_t StrideB,
                                                      std::array<index_t, 0> StrideDs, index_t StrideE,
                                                      PassThrough a_op, PassThrough b_op,
                                                      CDEElementwiseOperation cde_op) override
    {
        return std::make_unique<Argument>(
            MakeArgument(p_a, p_b, p_ds, p_e, M, N, K, StrideA, StrideB, StrideDs, StrideE, a_op, b_op, cde_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }
    // optional scratch for the planner's split-K of short grids (int32 partials): device_base.hpp:60-67 conventions, the
    // caller allocates and hands it over with SetWorkSpacePointer; without it the launch runs one slice
    size_t GetWorkSpaceSize(const BaseArgument* p_arg) const override
    {
        return wqgemm_workspace_bytes(&dynamic_cast<const Argument*>(p_arg)->prob);
    }
    std::string GetTypeString() const override { return "DeviceGemmQuantization_gfx950<I8, I8, I8, Activation_Mul_Clamp>"; }
};

// ---------------------------------------------------------------------------------------------
// E[M,N] = Add_Activation_Mul_Clamp( A[M,K] int8 * B[K,N] int8, bias[N] int32 )   int8 GEMM + bias + requantisation
// The call shape of example/14_gemm_quantization/gemm_xdl_bias_relu_quantization_int8.cpp:51-94,172-187: one D
// tensor, the int32 bias, row-major with stride 0 (one value per column).  Activation is PassThrough or Relu.
// ---------------------------------------------------------------------------------------------
template <typename ALayout, typename BLayout, typename ELayout, typename Activation>
struct DeviceGemmBiasQuantization_gfx950
    : public DeviceGemmMultipleD<ALayout, BLayout, ck::Tuple<tensor_layout::gemm::RowMajor>, ELayout, int8_t, int8_t,
                                 ck::Tuple<int32_t>, int8_t, element_wise::PassThrough, element_wise::PassThrough,
                                 element_wise::Add_Activation_Mul_Clamp<Activation>>
{
    static_assert(std::is_same_v<Activation, element_wise::PassThrough> || std::is_same_v<Activation, element_wise::Relu>,
                  "Activation must be PassThrough or Relu");
    using Argument                = wq_detail::Argument;
    using Invoker                 = wq_detail::Invoker;
    using PassThrough             = element_wise::PassThrough;
    using CDEElementwiseOperation = element_wise::Add_Activation_Mul_Clamp<Activation>;

    static Argument MakeArgument(const void* p_a, const void* p_b, std::array<const void*, 1> p_ds, void* p_e,
                                 index_t M, index_t N, index_t K, index_t StrideA, index_t StrideB,
                                 std::array<index_t, 1> StrideDs, index_t StrideE, PassThrough, PassThrough,
                                 CDEElementwiseOperation cde_op)
    {
        Argument arg;
        wqgemm_problem& p = arg.prob;
        p.a_dtype = p.b_dtype = p.c_dtype = WQ_I8;
        p.scale_dtype                     = WQ_F32;
        p.a_layout                        = wq_detail::layout_of<ALayout>();
        p.b_layout                        = wq_detail::layout_of<BLayout>();
        p.c_layout                        = wq_detail::layout_of<ELayout>();
        p.M = M, p.N = N, p.K = K;
        p.lda = StrideA, p.ldb = StrideB, p.ldc = StrideE;
        p.a = p_a, p.b = p_b, p.c = p_e;
        p.d0            = p_ds[0];
        p.epilogue      = std::is_same_v<Activation, element_wise::Relu> ? WQ_EPI_ADD_RELU_MUL_CLAMP : WQ_EPI_ADD_ACT_MUL_CLAMP;
        p.requant_scale = cde_op.requantScale_;
        p.k_batch       = 1;
        if(StrideDs[0] != 0) // a full (M, N) bias matrix is not a per-column bias: refused by IsSupportedArgument
            p.epilogue = -1;
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{"DeviceGemmBiasQuantization_gfx950"}; }
    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument> MakeArgumentPointer(const void* p_a, const void* p_b,
                                                      std::array<const void*, 1> p_ds, void* p_e, index_t M,
                                                      index_t N, index_t K, index_t StrideA, index_t StrideB,
                                                      std::array<index_t, 1> StrideDs, index_t StrideE,
                                                      PassThrough a_op, PassThrough b_op,
                                                      CDEElementwiseOperation cde_op) override
    {
        return std::make_unique<Argument>(
            MakeArgument(p_a, p_b, p_ds, p_e, M, N, K, StrideA, StrideB, StrideDs, StrideE, a_op, b_op, cde_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }
    // optional scratch for the planner's split-K of short grids (int32 partials): device_base.hpp:60-67 conventions, the
    // caller allocates and hands it over with SetWorkSpacePointer; without it the launch runs one slice
    size_t GetWorkSpaceSize(const BaseArgument* p_arg) const override
    {
        return wqgemm_workspace_bytes(&dynamic_cast<const Argument*>(p_arg)->prob);
    }
    std::string GetTypeString() const override
    {
        return "DeviceGemmBiasQuantization_gfx950<I8, I8, I32, I8, Add_Activation_Mul_Clamp>";
    }
};

// ---------------------------------------------------------------------------------------------
// E = cde_op( A0[M,K] bf16 * Multiply(B0[K,N] int8, B1[N] bf16) )                 multi-ABD form
// ---------------------------------------------------------------------------------------------
template <typename A0Layout, typename B0Layout, typename ELayout, typename A0DataType, typename B0DataType,
          typename B1DataType, typename EDataType, typename CDEElementwiseOperation>
struct DeviceGemmMultiABD_WeightQuant_gfx950
    : public DeviceGemmMultipleABD<ck::Tuple<A0Layout>, ck::Tuple<B0Layout, B0Layout>, ck::Tuple<>, ELayout,
                                   ck::Tuple<A0DataType>, ck::Tuple<B0DataType, B1DataType>, ck::Tuple<>,
                                   EDataType, element_wise::PassThrough, element_wise::Multiply,
                                   CDEElementwiseOperation>
{
    using Argument    = wq_detail::Argument;
    using Invoker     = wq_detail::Invoker;
    using PassThrough = element_wise::PassThrough;
    using Multiply    = element_wise::Multiply;
    static_assert(wq_detail::unary_epilogue_of<CDEElementwiseOperation>() >= 0,
                  "CDEElementwiseOperation must be PassThrough or FastGelu");

    static Argument MakeArgument(std::array<const void*, 1> p_as, std::array<const void*, 2> p_bs,
                                 std::array<const void*, 0>, void* p_e, index_t M, index_t N, index_t K,
                                 std::array<index_t, 1> StrideAs, std::array<index_t, 2> StrideBs,
                                 std::array<index_t, 0>, index_t StrideE, PassThrough, Multiply,
                                 CDEElementwiseOperation)
    {
        Argument arg;
        wqgemm_problem& p = arg.prob;
        p.a_dtype         = wq_detail::dtype_of<A0DataType>::value;
        p.b_dtype         = wq_detail::dtype_of<B0DataType>::value;
        p.scale_dtype     = wq_detail::dtype_of<B1DataType>::value;
        p.c_dtype         = wq_detail::dtype_of<EDataType>::value;
        p.a_layout        = wq_detail::layout_of<A0Layout>();
        p.b_layout        = wq_detail::layout_of<B0Layout>();
        p.c_layout        = wq_detail::layout_of<ELayout>();
        p.M = M, p.N = N, p.K = K;
        p.lda = StrideAs[0], p.ldb = StrideBs[0], p.ldc = StrideE; // StrideBs[1] == 0: per-column scale
        p.a = p_as[0], p.b = p_bs[0], p.scale = p_bs[1], p.c = p_e;
        p.epilogue = wq_detail::unary_epilogue_of<CDEElementwiseOperation>();
        if(StrideBs[1] != 0) // a scale per (k, n) is not a per-column scale: refused by IsSupportedArgument
            p.epilogue = -1;
        p.k_batch = 1;
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{"DeviceGemmMultiABD_WeightQuant_gfx950"}; }
    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument>
    MakeArgumentPointer(std::array<const void*, 1> p_as, std::array<const void*, 2> p_bs,
                        std::array<const void*, 0> p_ds, void* p_e, index_t M, index_t N, index_t K,
                        std::array<index_t, 1> StrideAs, std::array<index_t, 2> StrideBs,
                        std::array<index_t, 0> StrideDs, index_t StrideE, PassThrough a_op, Multiply b_op,
                        CDEElementwiseOperation cde_op) override
    {
        return std::make_unique<Argument>(MakeArgument(p_as, p_bs, p_ds, p_e, M, N, K, StrideAs, StrideBs,
                                                       StrideDs, StrideE, a_op, b_op, cde_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }
    std::string GetTypeString() const override { return "DeviceGemmMultiABD_WeightQuant_gfx950<BF16, I8, Multiply>"; }
};

// ---------------------------------------------------------------------------------------------
// The whole multi-ABD instance family of the reference for weight-quantized B
// (library/include/ck/library/tensor_operation_instance/gpu/gemm_multi_abd.hpp:24-260):
//   ScaleOnB = true : Bs = (B0 int8, B1 scale[N]), BOp = Multiply;   CDE = PassThrough | FastGelu | ConvertBF16RTN (no D),
//                                                                     Add | AddFastGelu (D0 = bias[N])
//   ScaleOnB = false: Bs = (B0 int8),              BOp = PassThrough; CDE = Multiply | MultiplyFastGelu
//                                                                     (D0 = scale[N]), MultiplyAddFastGelu | MultiplyAdd
//                                                                     (D0 = scale[N], D1 = bias[N])
// D tensors have lengths (M, N) and stride 0 along M: N values of E's type, or a row pitch for full matrices.
// (The reference's MultiplyAdd for bhalf_t, element_wise_operation.hpp:200-231, multiplies and adds the 16-bit patterns
// as integers; it is reproduced as written.)
// ---------------------------------------------------------------------------------------------
namespace wq_detail {
template <typename T, int N>
struct repeat_tuple;
template <typename T>
struct repeat_tuple<T, 0>
{
    using type = ck::Tuple<>;
};
template <typename T>
struct repeat_tuple<T, 1>
{
    using type = ck::Tuple<T>;
};
template <typename T>
struct repeat_tuple<T, 2>
{
    using type = ck::Tuple<T, T>;
};

template <bool ScaleOnB, typename CDE>
constexpr int multi_abd_epilogue()
{
    namespace ew = ck::tensor_operation::element_wise;
    if constexpr(ScaleOnB)
    {
        if constexpr(std::is_same_v<CDE, ew::PassThrough>)
            return WQ_EPI_PASSTHROUGH;
        else if constexpr(std::is_same_v<CDE, ew::FastGelu>)
            return WQ_EPI_FASTGELU;
        else if constexpr(std::is_same_v<CDE, ew::Add>)
            return WQ_EPI_ADD;
        else if constexpr(std::is_same_v<CDE, ew::AddFastGelu>)
            return WQ_EPI_ADD_FASTGELU;
        else if constexpr(std::is_same_v<CDE, ew::ConvertBF16RTN>)
            return WQ_EPI_CONVERT_BF16_RTN;
        else
            return -1;
    }
    else
    {
        if constexpr(std::is_same_v<CDE, ew::ConvertBF16RTN>) // unscaled int8 weights, nearest-even bf16 output
            return WQ_EPI_CONVERT_BF16_RTN;
        else if constexpr(std::is_same_v<CDE, ew::Multiply>)
            return WQ_EPI_MULTIPLY;
        else if constexpr(std::is_same_v<CDE, ew::MultiplyFastGelu>)
            return WQ_EPI_MULTIPLY_FASTGELU;
        else if constexpr(std::is_same_v<CDE, ew::MultiplyAddFastGelu>)
            return WQ_EPI_MULTIPLY_ADD_FASTGELU;
        else if constexpr(std::is_same_v<CDE, ew::MultiplyAdd>)
            return WQ_EPI_MULTIPLY_ADD;
        else
            return -1;
    }
}
constexpr int num_d_of(int epilogue)
{
    return epilogue == WQ_EPI_PASSTHROUGH || epilogue == WQ_EPI_FASTGELU || epilogue == WQ_EPI_CONVERT_BF16_RTN
               ? 0
               : (epilogue == WQ_EPI_MULTIPLY_ADD_FASTGELU || epilogue == WQ_EPI_MULTIPLY_ADD ? 2 : 1);
}
} // namespace wq_detail

template <typename A0Layout, typename B0Layout, typename ELayout, typename A0DataType, typename B0DataType,
          typename EDataType, bool ScaleOnB, typename CDEElementwiseOperation,
          int NumB = ScaleOnB ? 2 : 1,
          int NumD = wq_detail::num_d_of(wq_detail::multi_abd_epilogue<ScaleOnB, CDEElementwiseOperation>())>
struct DeviceGemmMultiABD_gfx950
    : public DeviceGemmMultipleABD<
          ck::Tuple<A0Layout>, typename wq_detail::repeat_tuple<B0Layout, NumB>::type,
          typename wq_detail::repeat_tuple<ck::tensor_layout::gemm::RowMajor, NumD>::type, ELayout, ck::Tuple<A0DataType>,
          std::conditional_t<ScaleOnB, ck::Tuple<B0DataType, A0DataType>, ck::Tuple<B0DataType>>,
          typename wq_detail::repeat_tuple<EDataType, NumD>::type, EDataType, element_wise::PassThrough,
          std::conditional_t<ScaleOnB, element_wise::Multiply, element_wise::PassThrough>, CDEElementwiseOperation>
{
    using Argument    = wq_detail::Argument;
    using Invoker     = wq_detail::Invoker;
    using PassThrough = element_wise::PassThrough;
    using BOp         = std::conditional_t<ScaleOnB, element_wise::Multiply, element_wise::PassThrough>;
    static constexpr int kEpilogue = wq_detail::multi_abd_epilogue<ScaleOnB, CDEElementwiseOperation>();
    static_assert(kEpilogue >= 0, "unsupported CDE element-wise operation for this B form");

    static Argument MakeArgument(std::array<const void*, 1> p_as, std::array<const void*, NumB> p_bs,
                                 std::array<const void*, NumD> p_ds, void* p_e, index_t M, index_t N, index_t K,
                                 std::array<index_t, 1> StrideAs, std::array<index_t, NumB> StrideBs,
                                 std::array<index_t, NumD> StrideDs, index_t StrideE, PassThrough, BOp,
                                 CDEElementwiseOperation)
    {
        Argument arg;
        wqgemm_problem& p = arg.prob;
        p.a_dtype         = wq_detail::dtype_of<A0DataType>::value;
        p.b_dtype         = wq_detail::dtype_of<B0DataType>::value;
        p.scale_dtype     = p.a_dtype;
        p.c_dtype         = wq_detail::dtype_of<EDataType>::value;
        p.a_layout        = wq_detail::layout_of<A0Layout>();
        p.b_layout        = wq_detail::layout_of<B0Layout>();
        p.c_layout        = wq_detail::layout_of<ELayout>();
        p.M = M, p.N = N, p.K = K;
        p.lda = StrideAs[0], p.ldb = StrideBs[0], p.ldc = StrideE;
        p.a = p_as[0], p.b = p_bs[0], p.c = p_e;
        p.epilogue = kEpilogue;
        if constexpr(ScaleOnB)
        {
            p.scale = p_bs[1]; // lengths (K, N), stride 0 along K: N values
            if(StrideBs[1] != 0) // a scale per (k, n) is not a per-column scale: refused by IsSupportedArgument
                p.epilogue = -1;
        }
        // D tensors are row-major (M, N): stride 0 = one value per column, otherwise the row pitch of a full matrix
        // (client_example/30_gemm_bf16Aint8B/gemm_bias_xdl_bf16_i8.cpp:84,128 hands over a full one)
        if constexpr(NumD >= 1)
            p.d0 = p_ds[0], p.ldd0 = StrideDs[0];
        if constexpr(NumD >= 2)
            p.d1 = p_ds[1], p.ldd1 = StrideDs[1];
        p.k_batch = 1;
        arg.AllocateOptionalWorkspace();
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{"DeviceGemmMultiABD_gfx950"}; }
    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument>
    MakeArgumentPointer(std::array<const void*, 1> p_as, std::array<const void*, NumB> p_bs,
                        std::array<const void*, NumD> p_ds, void* p_e, index_t M, index_t N, index_t K,
                        std::array<index_t, 1> StrideAs, std::array<index_t, NumB> StrideBs,
                        std::array<index_t, NumD> StrideDs, index_t StrideE, PassThrough a_op, BOp b_op,
                        CDEElementwiseOperation cde_op) override
    {
        return std::make_unique<Argument>(MakeArgument(p_as, p_bs, p_ds, p_e, M, N, K, StrideAs, StrideBs, StrideDs,
                                                       StrideE, a_op, b_op, cde_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }
    std::string GetTypeString() const override
    {
        return std::string("DeviceGemmMultiABD_gfx950<BF16, I8, ") + (ScaleOnB ? "MultiplyOnB" : "ScaleInEpilogue") +
               ", Epilogue" + std::to_string(kEpilogue) + ">";
    }
};

// ---------------------------------------------------------------------------------------------
// E = cde_op( A[M,K] fp8 * B[K,N] int4/int8, D0[N] f32, D1[M] f32 )       multiply_multiply form
// ---------------------------------------------------------------------------------------------
// ADataType ck::f8_t (E4M3 FNUZ, the reference's fp8) or ck::wq_f8_e4m3fn_t (OCP E4M3, native to the gfx950
// matrix core).  The matrix core cannot express three FNUZ encodings (0x7F / 0xFF = +-240 and 0x80 = NaN), so
// FNUZ data runs on it only with AssumeNoFnuzSpecials = true, by which the caller vouches that the
// activations hold none of them; otherwise FNUZ data goes through an fp16 image in the workspace (GetWorkSpaceSize)
// or, without a workspace, the exact general kernel.
template <typename ALayout, typename BLayout, typename ELayout, typename ADataType, typename BDataType,
          typename EDataType, typename CDEElementwiseOperation, bool AssumeNoFnuzSpecials = false>
struct DeviceGemmMultiplyMultiply_WeightQuant_gfx950
    : public DeviceGemmMultipleD<ALayout, BLayout,
                                 ck::Tuple<tensor_layout::gemm::RowMajor, tensor_layout::gemm::ColumnMajor>, ELayout,
                                 ADataType, BDataType, ck::Tuple<float, float>, EDataType, element_wise::PassThrough,
                                 element_wise::PassThrough, CDEElementwiseOperation>
{
    using Argument    = wq_detail::Argument;
    using Invoker     = wq_detail::Invoker;
    using PassThrough = element_wise::PassThrough;

    static constexpr int epilogue()
    {
        if constexpr(std::is_same_v<CDEElementwiseOperation, element_wise::MultiplyMultiply>)
            return WQ_EPI_MULTIPLY_MULTIPLY;
        else
            return WQ_EPI_MULTIPLY_MULTIPLY_FASTGELU;
    }

    // D0 row-major (M, N), D1 column-major (M, N); stride 0 = D0 per column, D1 per row, the form
    // example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8.cpp:43-47,152-153 uses; other strides address
    // full matrices
    static Argument MakeArgument(const void* p_a, const void* p_b, std::array<const void*, 2> p_ds, void* p_e,
                                 index_t M, index_t N, index_t K, index_t StrideA, index_t StrideB,
                                 std::array<index_t, 2> StrideDs, index_t StrideE, PassThrough, PassThrough,
                                 CDEElementwiseOperation)
    {
        Argument arg;
        wqgemm_problem& p = arg.prob;
        p.a_dtype         = wq_detail::dtype_of<ADataType>::value;
        p.b_dtype         = wq_detail::dtype_of<BDataType>::value;
        p.scale_dtype     = WQ_F32;
        p.c_dtype         = wq_detail::dtype_of<EDataType>::value;
        p.a_layout        = wq_detail::layout_of<ALayout>();
        p.b_layout        = wq_detail::layout_of<BLayout>();
        p.c_layout        = wq_detail::layout_of<ELayout>();
        p.M = M, p.N = N, p.K = K;
        p.lda = StrideA, p.ldb = StrideB, p.ldc = StrideE;
        p.a = p_a, p.b = p_b, p.c = p_e;
        p.d0 = p_ds[0], p.d1 = p_ds[1];
        p.ldd0 = StrideDs[0], p.ldd1 = StrideDs[1];
        p.epilogue = epilogue();
        p.k_batch  = 1;
        p.flags    = AssumeNoFnuzSpecials ? WQ_FLAG_F8_NO_SPECIALS : 0;
        arg.AllocateOptionalWorkspace();
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{"DeviceGemmMultiplyMultiply_WeightQuant_gfx950"}; }
    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument> MakeArgumentPointer(const void* p_a, const void* p_b,
                                                      std::array<const void*, 2> p_ds, void* p_e, index_t M,
                                                      index_t N, index_t K, index_t StrideA, index_t StrideB,
                                                      std::array<index_t, 2> StrideDs, index_t StrideE,
                                                      PassThrough a_op, PassThrough b_op,
                                                      CDEElementwiseOperation cde_op) override
    {
        return std::make_unique<Argument>(
            MakeArgument(p_a, p_b, p_ds, p_e, M, N, K, StrideA, StrideB, StrideDs, StrideE, a_op, b_op, cde_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }
    // (M*K + N)*2 bytes for FNUZ activations that are not vouched free of the special encodings: room for the
    // fp16 image of A (device_base.hpp:60-67 conventions: caller allocates, SetWorkSpacePointer hands it over)
    size_t GetWorkSpaceSize(const BaseArgument* p_arg) const override
    {
        return wqgemm_workspace_bytes(&dynamic_cast<const Argument*>(p_arg)->prob);
    }
    std::string GetTypeString() const override { return "DeviceGemmMultiplyMultiply_WeightQuant_gfx950<F8, I4x2>"; }
};

// ---------------------------------------------------------------------------------------------
// The same GEMM for int4 weights held ONE PER BYTE -- the reference's int4 convention: ck::int4_t = _BitInt(4) occupies a
// byte (include/ck/utility/data_type.hpp:10-14), and its int4 examples upload int8 tensors holding -8 .. 7
// (example/01_gemm/run_gemm_example.inc:93-95,181-191).  BValueType is any one-byte type whose low four bits are the
// two's-complement value (ck::int4_t, or int8_t at the caller's word that the values lie in -8 .. 7).  Every Run packs the
// weights two per byte into argument-owned memory (wqgemm_pack_int4, K * N / 2 bytes, untimed -- in the manner of
// launch_and_time_kernel_with_preprocess, include/ck/host_utility/flush_cache.hpp:107-235) and then runs the operator
// above on the packed weights, so CK's own callers -- its ckProfiler loop, profile_gemm_multiply_multiply_impl.hpp --
// reach the fp8 x int4 kernels with tensors generated, uploaded and verified by CK's code unchanged.  Column-major B
// (K contiguous), K a multiple of 8.
// ---------------------------------------------------------------------------------------------
template <typename ALayout, typename BLayout, typename ELayout, typename ADataType, typename BValueType,
          typename EDataType, typename CDEElementwiseOperation, bool AssumeNoFnuzSpecials = false>
struct DeviceGemmMultiplyMultiply_Int4Values_gfx950
    : public DeviceGemmMultipleD<ALayout, BLayout,
                                 ck::Tuple<tensor_layout::gemm::RowMajor, tensor_layout::gemm::ColumnMajor>, ELayout,
                                 ADataType, BValueType, ck::Tuple<float, float>, EDataType, element_wise::PassThrough,
                                 element_wise::PassThrough, CDEElementwiseOperation>
{
    static_assert(sizeof(BValueType) == 1, "one int4 value per byte");
    using Packed      = DeviceGemmMultiplyMultiply_WeightQuant_gfx950<ALayout, BLayout, ELayout, ADataType, ck::wq_int4x2_t,
                                                                      EDataType, CDEElementwiseOperation, AssumeNoFnuzSpecials>;
    using PassThrough = element_wise::PassThrough;

    struct Argument : public wq_detail::Argument
    {
        const void* b_values = nullptr; // the caller's weights, one per byte
        index_t ld_values    = 0;
        std::shared_ptr<void> packed;   // K / 2 bytes per column
        bool packable        = false;
    };

    struct Invoker : public BaseInvoker
    {
        float Run(const Argument& arg, const StreamConfig& cfg = StreamConfig{})
        {
            if(!arg.packable || !arg.packed)
                throw std::runtime_error("wrong! DeviceGemmMultiplyMultiply_Int4Values_gfx950 has invalid setting: column-major B "
                                         "with K a multiple of 8 and StrideB >= K, and K * N / 2 bytes of device memory");
            const wqgemm_problem& p = arg.prob;
            if(wqgemm_pack_int4((const int8_t*)arg.b_values, arg.ld_values, p.K, p.N, (uint8_t*)arg.packed.get(), p.K / 2,
                                cfg.stream_id_) != WQ_OK)
                throw std::runtime_error(std::string("DeviceGemmMultiplyMultiply_Int4Values_gfx950: ") + wqgemm_last_error());
            return wq_detail::run(p, arg.Workspace(), cfg, "DeviceGemmMultiplyMultiply_Int4Values_gfx950");
        }
        float Run(const BaseArgument* p_arg, const StreamConfig& cfg = StreamConfig{}) override
        {
            return Run(*dynamic_cast<const Argument*>(p_arg), cfg);
        }
    };

    static Argument MakeArgument(const void* p_a, const void* p_b, std::array<const void*, 2> p_ds, void* p_e, index_t M,
                                 index_t N, index_t K, index_t StrideA, index_t StrideB, std::array<index_t, 2> StrideDs,
                                 index_t StrideE, PassThrough, PassThrough, CDEElementwiseOperation cde)
    {
        Argument arg;
        arg.b_values  = p_b;
        arg.ld_values = StrideB;
        arg.packable  = std::is_same_v<BLayout, tensor_layout::gemm::ColumnMajor> && K > 0 && N > 0 && K % 8 == 0 && StrideB >= K;
        if(arg.packable)
        {
            arg.packed = wq_detail::ScratchPool::instance().take((size_t)N * (K / 2));
        }
        // the packed problem: same operands, B = the packed copy with K / 2 bytes per column
        wq_detail::Argument inner = Packed::MakeArgument(p_a, arg.packed.get(), p_ds, p_e, M, N, K, StrideA, K, StrideDs, StrideE,
                                                         PassThrough{}, PassThrough{}, cde);
        arg.prob            = inner.prob;
        arg.owned_workspace = inner.owned_workspace;
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{}; }
    static bool IsSupportedArgument(const Argument& arg)
    {
        return arg.packable && arg.packed && wq_detail::supported(arg.prob);
    }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument> MakeArgumentPointer(const void* p_a, const void* p_b, std::array<const void*, 2> p_ds,
                                                      void* p_e, index_t M, index_t N, index_t K, index_t StrideA,
                                                      index_t StrideB, std::array<index_t, 2> StrideDs, index_t StrideE,
                                                      PassThrough a_op, PassThrough b_op,
                                                      CDEElementwiseOperation cde_op) override
    {
        return std::make_unique<Argument>(
            MakeArgument(p_a, p_b, p_ds, p_e, M, N, K, StrideA, StrideB, StrideDs, StrideE, a_op, b_op, cde_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(); }
    size_t GetWorkSpaceSize(const BaseArgument* p_arg) const override
    {
        return wqgemm_workspace_bytes(&dynamic_cast<const Argument*>(p_arg)->prob);
    }
    std::string GetTypeString() const override
    {
        return std::string("DeviceGemmMultiplyMultiply_Int4Values_gfx950<") +
               (std::is_same_v<ADataType, ck::wq_f8_e4m3fn_t> ? "F8_E4M3FN" : AssumeNoFnuzSpecials ? "F8_FNUZ_NoSpecials" : "F8_FNUZ") +
               ", int4 per byte -> I4x2 packed per Run>";
    }
};

// ---------------------------------------------------------------------------------------------
// E[M,N] = convert( sum_k (float(A[m,k]) * a_scale[m / ScaleBlockM][k / ScaleBlockK])
//                        * (float(B[k,n]) * b_scale[k / ScaleBlockK][n / ScaleBlockN]) )
// fp8 activations, int4 (packed / pre-shuffled) or int8 weights, fp32 block scales of BOTH operands -- the problem
// of example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:58-75,126-152,262-283 with
// weight-quantized B.  a_scale is row-major (ceil(M/ScaleBlockM), ceil(K/ScaleBlockK)), b_scale column-major
// (ceil(K/ScaleBlockK), ceil(N/ScaleBlockN)), both densely packed (leading dimension ceil(K/ScaleBlockK), as the
// example's Scale_Stride_AM / Scale_Stride_BN).  With 128-k blocks (ScaleBlockM a multiple of 128, ScaleBlockN of 32)
// the fp8 matrix-core kernel scales each 128-k block sum in registers; other block shapes run on the general kernel.
// ---------------------------------------------------------------------------------------------
template <typename ALayout, typename BLayout, typename ELayout, typename ADataType, typename BDataType,
          typename EDataType, index_t ScaleBlockM, index_t ScaleBlockN, index_t ScaleBlockK,
          bool AssumeNoFnuzSpecials = false>
struct DeviceGemmMultiD_ABScale_WeightQuant_gfx950
    : public DeviceGemmMultipleD_ABScale<ALayout, BLayout, ck::Tuple<>, ELayout, ADataType, float, BDataType, float,
                                         ck::Tuple<>, EDataType, ScaleBlockM, ScaleBlockN, ScaleBlockK,
                                         element_wise::PassThrough, element_wise::PassThrough, element_wise::PassThrough>
{
    using Argument    = wq_detail::Argument;
    using Invoker     = wq_detail::Invoker;
    using PassThrough = element_wise::PassThrough;

    static Argument MakeArgument(const void* p_a, const void* p_b, std::array<const void*, 0>, void* p_e, index_t M,
                                 index_t N, index_t K, index_t StrideA, index_t StrideB, std::array<index_t, 0>,
                                 index_t StrideE, const void* p_a_scale, const void* p_b_scale, PassThrough,
                                 PassThrough, PassThrough)
    {
        Argument arg;
        wqgemm_problem& p = arg.prob;
        p.a_dtype         = wq_detail::dtype_of<ADataType>::value;
        p.b_dtype         = wq_detail::dtype_of<BDataType>::value;
        p.scale_dtype     = WQ_F32;
        p.c_dtype         = wq_detail::dtype_of<EDataType>::value;
        p.a_layout        = wq_detail::layout_of<ALayout>();
        p.b_layout        = wq_detail::layout_of<BLayout>();
        p.c_layout        = wq_detail::layout_of<ELayout>();
        p.M = M, p.N = N, p.K = K;
        p.lda = StrideA, p.ldb = StrideB, p.ldc = StrideE;
        p.a = p_a, p.b = p_b, p.c = p_e;
        p.epilogue      = WQ_EPI_PASSTHROUGH;
        p.k_batch       = 1;
        p.flags         = AssumeNoFnuzSpecials ? WQ_FLAG_F8_NO_SPECIALS : 0;
        p.a_block_scale = p_a_scale, p.b_block_scale = p_b_scale;
        p.scale_block_m = ScaleBlockM, p.scale_block_n = ScaleBlockN, p.scale_block_k = ScaleBlockK;
        p.ld_a_scale = p.ld_b_scale = 0; // densely packed
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{"DeviceGemmMultiD_ABScale_WeightQuant_gfx950"}; }
    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument> MakeArgumentPointer(const void* p_a, const void* p_b, std::array<const void*, 0> p_ds,
                                                      void* p_e, const index_t M, const index_t N, const index_t K,
                                                      const index_t StrideA, const index_t StrideB,
                                                      const std::array<index_t, 0> StrideDs, const index_t StrideE,
                                                      const void* p_a_scale, const void* p_b_scale, PassThrough a_op,
                                                      PassThrough b_op, PassThrough cde_op) override
    {
        return std::make_unique<Argument>(MakeArgument(p_a, p_b, p_ds, p_e, M, N, K, StrideA, StrideB, StrideDs, StrideE,
                                                       p_a_scale, p_b_scale, a_op, b_op, cde_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }
    std::string GetTypeString() const override
    {
        return std::string("DeviceGemmMultiD_ABScale_WeightQuant_gfx950<") +
               (std::is_same_v<ADataType, ck::wq_f8_e4m3fn_t> ? "F8_E4M3FN" : AssumeNoFnuzSpecials ? "F8_FNUZ_NoSpecials" : "F8_FNUZ") +
               ", " + std::to_string(ScaleBlockM) + "x" + std::to_string(ScaleBlockN) + "x" + std::to_string(ScaleBlockK) + ">";
    }
};

// ---------------------------------------------------------------------------------------------
// The block-scaled GEMM for int4 weights held ONE PER BYTE (ck::int4_t, the reference's int4 convention; see
// DeviceGemmMultiplyMultiply_Int4Values_gfx950 above): every Run packs the weights two per byte into pooled memory
// (untimed) and runs DeviceGemmMultiD_ABScale_WeightQuant_gfx950 on the packed copy, so that CK's own profiling loop
// for this interface (profiler/include/profiler/profile_gemm_ab_scale_impl.hpp) drives the kernels with tensors it
// generated, uploaded and verifies itself.  Column-major B (K contiguous), K a multiple of 8.
// ---------------------------------------------------------------------------------------------
template <typename ALayout, typename BLayout, typename ELayout, typename ADataType, typename BValueType,
          typename EDataType, index_t ScaleBlockM, index_t ScaleBlockN, index_t ScaleBlockK,
          bool AssumeNoFnuzSpecials = false>
struct DeviceGemmMultiD_ABScale_Int4Values_gfx950
    : public DeviceGemmMultipleD_ABScale<ALayout, BLayout, ck::Tuple<>, ELayout, ADataType, float, BValueType, float,
                                         ck::Tuple<>, EDataType, ScaleBlockM, ScaleBlockN, ScaleBlockK,
                                         element_wise::PassThrough, element_wise::PassThrough, element_wise::PassThrough>
{
    static_assert(sizeof(BValueType) == 1, "one int4 value per byte");
    using Packed = DeviceGemmMultiD_ABScale_WeightQuant_gfx950<ALayout, BLayout, ELayout, ADataType, ck::wq_int4x2_t, EDataType,
                                                               ScaleBlockM, ScaleBlockN, ScaleBlockK, AssumeNoFnuzSpecials>;
    using PassThrough = element_wise::PassThrough;

    struct Argument : public wq_detail::Argument
    {
        const void* b_values = nullptr; // the caller's weights, one per byte
        index_t ld_values    = 0;
        std::shared_ptr<void> packed;   // K / 2 bytes per column
        bool packable        = false;
    };

    struct Invoker : public BaseInvoker
    {
        float Run(const Argument& arg, const StreamConfig& cfg = StreamConfig{})
        {
            if(!arg.packable || !arg.packed)
                throw std::runtime_error("wrong! DeviceGemmMultiD_ABScale_Int4Values_gfx950 has invalid setting: column-major B "
                                         "with K a multiple of 8 and StrideB >= K, and K * N / 2 bytes of device memory");
            const wqgemm_problem& p = arg.prob;
            if(wqgemm_pack_int4((const int8_t*)arg.b_values, arg.ld_values, p.K, p.N, (uint8_t*)arg.packed.get(), p.K / 2,
                                cfg.stream_id_) != WQ_OK)
                throw std::runtime_error(std::string("DeviceGemmMultiD_ABScale_Int4Values_gfx950: ") + wqgemm_last_error());
            return wq_detail::run(p, arg.Workspace(), cfg, "DeviceGemmMultiD_ABScale_Int4Values_gfx950");
        }
        float Run(const BaseArgument* p_arg, const StreamConfig& cfg = StreamConfig{}) override
        {
            return Run(*dynamic_cast<const Argument*>(p_arg), cfg);
        }
    };

    static Argument MakeArgument(const void* p_a, const void* p_b, std::array<const void*, 0> p_ds, void* p_e, index_t M,
                                 index_t N, index_t K, index_t StrideA, index_t StrideB, std::array<index_t, 0> StrideDs,
                                 index_t StrideE, const void* p_a_scale, const void* p_b_scale, PassThrough, PassThrough,
                                 PassThrough)
    {
        Argument arg;
        arg.b_values  = p_b;
        arg.ld_values = StrideB;
        arg.packable  = std::is_same_v<BLayout, tensor_layout::gemm::ColumnMajor> && K > 0 && N > 0 && K % 8 == 0 && StrideB >= K;
        if(arg.packable)
            arg.packed = wq_detail::ScratchPool::instance().take((size_t)N * (K / 2));
        wq_detail::Argument inner = Packed::MakeArgument(p_a, arg.packed.get(), p_ds, p_e, M, N, K, StrideA, K, StrideDs, StrideE,
                                                         p_a_scale, p_b_scale, PassThrough{}, PassThrough{}, PassThrough{});
        arg.prob = inner.prob;
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{}; }
    static bool IsSupportedArgument(const Argument& arg)
    {
        return arg.packable && arg.packed && wq_detail::supported(arg.prob);
    }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument> MakeArgumentPointer(const void* p_a, const void* p_b, std::array<const void*, 0> p_ds,
                                                      void* p_e, const index_t M, const index_t N, const index_t K,
                                                      const index_t StrideA, const index_t StrideB,
                                                      const std::array<index_t, 0> StrideDs, const index_t StrideE,
                                                      const void* p_a_scale, const void* p_b_scale, PassThrough a_op,
                                                      PassThrough b_op, PassThrough cde_op) override
    {
        return std::make_unique<Argument>(MakeArgument(p_a, p_b, p_ds, p_e, M, N, K, StrideA, StrideB, StrideDs, StrideE,
                                                       p_a_scale, p_b_scale, a_op, b_op, cde_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(); }
    std::string GetTypeString() const override
    {
        return std::string("DeviceGemmMultiD_ABScale_Int4Values_gfx950<") +
               (std::is_same_v<ADataType, ck::wq_f8_e4m3fn_t> ? "F8_E4M3FN" : AssumeNoFnuzSpecials ? "F8_FNUZ_NoSpecials" : "F8_FNUZ") +
               ", int4 per byte -> I4x2 packed per Run, " + std::to_string(ScaleBlockM) + "x" + std::to_string(ScaleBlockN) + "x" +
               std::to_string(ScaleBlockK) + ">";
    }
};

// ---------------------------------------------------------------------------------------------
// Split-K over weight-quantized B:  C = c_op( sum over KBatch slices of A[M,K] * cvt_A(B[K,N]) [, D0] )
// A fp16 / bf16, B int8 (or offset uint8 / packed int4) with no weight scale -- ReferenceGemm's arithmetic with
// BDataType = int8 (library/include/ck/library/reference_tensor_operation/cpu/reference_gemm.hpp:60-108), the problem of
// example/35_splitK_gemm/gemm_xdl_splitk_reduce_bf16A_i8B.cpp.  Slice z covers [z KRead, min(K, (z+1) KRead)),
// KRead = ceil(K / (KBatch * 64)) * 64 (gridwise_gemm_xdl_cshuffle_v3.hpp:192-197), fp32 slabs [KBatch][M][N] are
// summed in ascending order and converted (device_gemm_xdl_cshuffle_v3r1.hpp:209-270).  KBatch = 1 runs one pass
// with no slabs.
// ---------------------------------------------------------------------------------------------
namespace wq_detail {
template <typename COp, int NumD>
constexpr int splitk_epilogue()
{
    namespace ew = ck::tensor_operation::element_wise;
    if constexpr(NumD == 0)
        return unary_epilogue_of<COp>();
    else if constexpr(NumD == 1)
    {
        if constexpr(std::is_same_v<COp, ew::Add>)
            return WQ_EPI_ADD;
        else if constexpr(std::is_same_v<COp, ew::AddFastGelu>)
            return WQ_EPI_ADD_FASTGELU;
        else if constexpr(std::is_same_v<COp, ew::Multiply>)
            return WQ_EPI_MULTIPLY;
        else if constexpr(std::is_same_v<COp, ew::MultiplyFastGelu>)
            return WQ_EPI_MULTIPLY_FASTGELU;
        else
            return -1;
    }
    else
        return -1;
}

inline void fill_unscaled(wqgemm_problem& p, int a_dtype, int b_dtype, int c_dtype, int a_layout, int b_layout, int c_layout,
                          const void* p_a, const void* p_b, void* p_c, index_t M, index_t N, index_t K, index_t StrideA,
                          index_t StrideB, index_t StrideC, index_t KBatch, int epilogue)
{
    p.a_dtype = a_dtype, p.b_dtype = b_dtype, p.scale_dtype = a_dtype, p.c_dtype = c_dtype;
    p.a_layout = a_layout, p.b_layout = b_layout, p.c_layout = c_layout;
    p.M = M, p.N = N, p.K = K;
    p.lda = StrideA, p.ldb = StrideB, p.ldc = StrideC;
    p.a = p_a, p.b = p_b, p.scale = nullptr, p.c = p_c;
    p.epilogue = epilogue;
    p.k_batch  = KBatch < 1 ? 1 : KBatch;
}
} // namespace wq_detail

template <typename ALayout, typename BLayout, typename DsLayout, typename CLayout, typename ADataType,
          typename BDataType, typename DsDataType, typename CDataType, typename AElementwiseOperation,
          typename BElementwiseOperation, typename CElementwiseOperation>
struct DeviceGemmV2R1_WeightQuant_gfx950
    : public DeviceGemmV2R1<ALayout, BLayout, DsLayout, CLayout, ADataType, BDataType, DsDataType, CDataType,
                            AElementwiseOperation, BElementwiseOperation, CElementwiseOperation>
{
    using Argument = wq_detail::Argument;
    using Invoker  = wq_detail::Invoker;
    static constexpr index_t NumDTensor = DsDataType::Size();
    static constexpr int kEpilogue      = wq_detail::splitk_epilogue<CElementwiseOperation, NumDTensor>();
    static_assert(std::is_same_v<AElementwiseOperation, element_wise::PassThrough> &&
                      std::is_same_v<BElementwiseOperation, element_wise::PassThrough>,
                  "A and B element-wise operations must be PassThrough");
    static_assert(kEpilogue >= 0, "unsupported C element-wise operation / D tensor count");

    static Argument MakeArgument(const ADataType* p_a, const BDataType* p_b, std::array<const void*, NumDTensor> p_ds,
                                 CDataType* p_c, index_t M, index_t N, index_t K, index_t StrideA, index_t StrideB,
                                 std::array<index_t, NumDTensor> StrideDs, index_t StrideC, index_t KBatch,
                                 AElementwiseOperation, BElementwiseOperation, CElementwiseOperation)
    {
        Argument arg;
        wq_detail::fill_unscaled(arg.prob, wq_detail::dtype_of<ADataType>::value, wq_detail::dtype_of<BDataType>::value,
                                 wq_detail::dtype_of<CDataType>::value, wq_detail::layout_of<ALayout>(),
                                 wq_detail::layout_of<BLayout>(), wq_detail::layout_of<CLayout>(), p_a, p_b, p_c, M, N, K,
                                 StrideA, StrideB, StrideC, KBatch, kEpilogue);
        if constexpr(NumDTensor == 1)
            arg.prob.d0 = p_ds[0], arg.prob.ldd0 = StrideDs[0]; // row-major (M, N) like C; stride 0 = one value per column
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{"DeviceGemmV2R1_WeightQuant_gfx950"}; }
    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument>
    MakeArgumentPointer(const void* p_a, const void* p_b, std::array<const void*, NumDTensor> p_ds, void* p_c, index_t M,
                        index_t N, index_t K, index_t StrideA, index_t StrideB, std::array<index_t, NumDTensor> StrideDs,
                        index_t StrideC, index_t KSplit, AElementwiseOperation a_op, BElementwiseOperation b_op,
                        CElementwiseOperation c_op) override
    {
        return std::make_unique<Argument>(MakeArgument(static_cast<const ADataType*>(p_a), static_cast<const BDataType*>(p_b),
                                                       p_ds, static_cast<CDataType*>(p_c), M, N, K, StrideA, StrideB, StrideDs,
                                                       StrideC, KSplit, a_op, b_op, c_op));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }
    // KBatch * M * N * 4 bytes of fp32 slabs for KBatch > 1 (device_gemm_xdl_cshuffle_v3r1.hpp:686-694), plus K * N for
    // the K-contiguous copy of row-major 8-bit weights; the caller allocates and hands it over
    size_t GetWorkSpaceSize(const BaseArgument* p_arg) const override
    {
        return wqgemm_workspace_bytes(&dynamic_cast<const Argument*>(p_arg)->prob);
    }
    std::string GetTypeString() const override
    {
        return std::string("DeviceGemmV2R1_WeightQuant_gfx950<") + ALayout::name[0] + BLayout::name[0] + CLayout::name[0] +
               ", " + (wq_detail::dtype_of<ADataType>::value == WQ_BF16 ? "BF16" : "F16") + ", " +
               (wq_detail::dtype_of<BDataType>::value == WQ_I4_PACKED ? "I4x2" : "I8") + ", SplitKReduce, Epilogue" +
               std::to_string(kEpilogue) + ">";
    }
};

template <typename ALayout, typename BLayout, typename CLayout, typename ADataType, typename BDataType,
          typename CDataType, typename CElementwiseOperation, typename ComputeType = CDataType>
struct DeviceGemmSplitK_WeightQuant_gfx950
    : public DeviceGemmSplitK<ALayout, BLayout, CLayout, ADataType, BDataType, CDataType, element_wise::PassThrough,
                              element_wise::PassThrough, CElementwiseOperation, ComputeType>
{
    using Argument    = wq_detail::Argument;
    using Invoker     = wq_detail::Invoker;
    using PassThrough = element_wise::PassThrough;
    static constexpr int kEpilogue = wq_detail::unary_epilogue_of<CElementwiseOperation>();
    static_assert(kEpilogue >= 0, "CElementwiseOperation must be PassThrough or FastGelu");

    // The interface has no workspace call (its reference instances accumulate with atomics into C:
    // device_gemm_xdl_splitk_c_shuffle.hpp), so the argument owns the fp32 slabs.
    static Argument MakeArgument(const ADataType* p_a, const BDataType* p_b, CDataType* p_c, index_t M, index_t N,
                                 index_t K, index_t StrideA, index_t StrideB, index_t StrideC, PassThrough, PassThrough,
                                 CElementwiseOperation, index_t KBatch)
    {
        Argument arg;
        wq_detail::fill_unscaled(arg.prob, wq_detail::dtype_of<ADataType>::value, wq_detail::dtype_of<BDataType>::value,
                                 wq_detail::dtype_of<CDataType>::value, wq_detail::layout_of<ALayout>(),
                                 wq_detail::layout_of<BLayout>(), wq_detail::layout_of<CLayout>(), p_a, p_b, p_c, M, N, K,
                                 StrideA, StrideB, StrideC, KBatch, kEpilogue);
        arg.AllocateOptionalWorkspace(/*with_splitk_slabs=*/true);
        return arg;
    }
    static Invoker MakeInvoker() { return Invoker{"DeviceGemmSplitK_WeightQuant_gfx950"}; }
    static bool IsSupportedArgument(const Argument& arg) { return wq_detail::supported(arg.prob); }
    bool IsSupportedArgument(const BaseArgument* p_arg) override
    {
        return IsSupportedArgument(*dynamic_cast<const Argument*>(p_arg));
    }
    std::unique_ptr<BaseArgument> MakeArgumentPointer(const void* p_a, const void* p_b, void* p_c, index_t M, index_t N,
                                                      index_t K, index_t StrideA, index_t StrideB, index_t StrideC,
                                                      PassThrough a_op, PassThrough b_op, CElementwiseOperation c_op,
                                                      index_t KBatch) override
    {
        return std::make_unique<Argument>(MakeArgument(static_cast<const ADataType*>(p_a), static_cast<const BDataType*>(p_b),
                                                       static_cast<CDataType*>(p_c), M, N, K, StrideA, StrideB, StrideC, a_op,
                                                       b_op, c_op, KBatch));
    }
    std::unique_ptr<BaseInvoker> MakeInvokerPointer() override { return std::make_unique<Invoker>(MakeInvoker()); }
    std::string GetTypeString() const override
    {
        return std::string("DeviceGemmSplitK_WeightQuant_gfx950<") + ALayout::name[0] + BLayout::name[0] + CLayout::name[0] +
               ", " + (wq_detail::dtype_of<ADataType>::value == WQ_BF16 ? "BF16" : "F16") + ", " +
               (wq_detail::dtype_of<BDataType>::value == WQ_I4_PACKED ? "I4x2" : "I8") + ", SlabReduce>";
    }
};

} // namespace device
} // namespace tensor_operation
} // namespace ck
