// Command-line profiler for the gfx950 weight-quantized GEMM operators, in the shape of Composable Kernel's
// ckProfiler: an operation registry keyed by argv[1] (profiler/src/profiler.cpp:15-33,
// profiler_operation_registry.hpp:12-70), per-operation argv positions, "Perf:" / "Best Perf ..." lines and
// host-reference verification taken from
//   gemm_multiply_multiply   profiler/src/profile_gemm_multiply_multiply.cpp:36-93, .../profile_gemm_multiply_multiply_impl.hpp:131-330
//   gemm_ab_scale            profiler/src/profile_gemm_ab_scale.cpp:41-182, .../profile_gemm_ab_scale_impl.hpp:39-363
//   gemm_splitk              profiler/src/profile_gemm_splitk.cpp:34-75, .../profile_gemm_splitk_impl.hpp:101-285
//   gemm_quantization        the int8 row of profile_gemm.cpp:20-27 with example/14's requantisation epilogue
// The data-type enums are extended for the weight-quantized forms this engine adds (listed in each usage text).
// A stride of -1 (script/profile_mixed_gemm.sh:17-51) selects the packed default.
#include <cmath>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <random>
#include <string>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"
#include "ck/library/utility/check_err.hpp"
#include "ck/library/utility/device_memory.hpp"
#include "ck/library/utility/fill.hpp"
#include "ck/library/utility/host_tensor.hpp"
#include "ck/library/utility/host_tensor_generator.hpp"
#include "ck/library/reference_tensor_operation/cpu/reference_gemm.hpp"
#include "ck/library/reference_tensor_operation/cpu/reference_fpAintB_gemm.hpp"

namespace {

using Row         = ck::tensor_layout::gemm::RowMajor;
using Col         = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
namespace dev     = ck::tensor_operation::device;

using Operation = std::function<int(int, char*[])>;
std::map<std::string, std::pair<std::string, Operation>>& registry()
{
    static std::map<std::string, std::pair<std::string, Operation>> r;
    return r;
}
struct Register
{
    Register(const char* name, const char* desc, Operation op) { registry()[name] = {desc, std::move(op)}; }
};

struct Common
{
    int data_type, layout;
    bool verify, log, time;
    int init;
    ck::index_t M, N, K;
};

// The reference's result block: one "Perf:" line per instance, then the best one with the problem.
void report(const char* dtype, ck::index_t M, ck::index_t N, ck::index_t K, ck::index_t sa, ck::index_t sb, ck::index_t se,
            float ms, std::size_t bytes, const std::string& name, const char* extra = "")
{
    const float tflops = 2.f * M * N * K / 1.E9f / ms, gbs = bytes / 1.E6f / ms;
    std::cout << "Perf: " << std::setw(10) << ms << " ms, " << tflops << " TFlops, " << gbs << " GB/s, " << name << extra
              << std::endl;
    std::cout << "Best Perf for datatype = " << dtype << " ALayout =  RowMajor BLayout =  ColumnMajor M = " << M
              << " N = " << N << " K = " << K << " StrideA = " << sa << " StrideB = " << sb << " StrideE = " << se << " : "
              << ms << " ms, " << tflops << " TFlops, " << gbs << " GB/s, " << name << extra << std::endl;
}

StreamConfig stream_config(bool time_kernel, int n_warmup, int n_iter, int rotating_count = 1)
{
    StreamConfig cfg{nullptr, time_kernel};
    cfg.cold_niters_   = n_warmup;
    cfg.nrepeat_       = n_iter;
    cfg.flush_cache    = rotating_count > 1; // as profile_gemm_multiply_multiply_impl.hpp:238 sets it
    cfg.rotating_count = rotating_count;
    return cfg;
}

// rotating_count from the "memory for rotating buffer" argument, as the reference derives it
// (profiler/include/profiler/profile_gemm_multiply_multiply_impl.hpp:78-84): enough copies of the inputs to fill that many
// bytes, at most one per timed iteration
int rotating_count_for(uint64_t rotating_bytes, std::size_t input_bytes, int n_iter)
{
    const int copies = static_cast<int>(std::ceil(static_cast<double>(rotating_bytes) / static_cast<double>(input_bytes)));
    return std::max(1, std::min(n_iter, copies));
}

bool parse_common(int argc, char* argv[], int first_fixed, Common& c)
{
    if(argc < first_fixed)
        return false;
    c.data_type = std::stoi(argv[2]);
    c.layout    = std::stoi(argv[3]);
    c.verify    = std::stoi(argv[4]);
    c.init      = std::stoi(argv[5]);
    c.log       = std::stoi(argv[6]);
    c.time      = std::stoi(argv[7]);
    c.M = std::stoi(argv[8]), c.N = std::stoi(argv[9]), c.K = std::stoi(argv[10]);
    if(c.layout != 1)
    {
        std::cerr << "this engine keeps weights K-contiguous: matrix layout must be 1 (A[m, k] * B[n, k] = C[m, n])\n";
        return false;
    }
    return true;
}

// ---- fp16 / bf16 activations x int4 / int8 weights, per-channel scale (+ split-K) ------------------------------
template <typename ADataType, bool Int4>
int run_fpaintb(const Common& c, ck::index_t sa, ck::index_t sb, ck::index_t sc, int kbatch, int n_warmup, int n_iter)
{
    using BDataType = std::conditional_t<Int4, ck::wq_int4x2_t, int8_t>;
    using Op = dev::DeviceFpAintBGemm_Mfma_gfx950<Row, Col, Row, ADataType, BDataType, ADataType, ADataType, PassThrough,
                                                  PassThrough, PassThrough>;
    const auto m = static_cast<std::size_t>(c.M), n = static_cast<std::size_t>(c.N), k = static_cast<std::size_t>(c.K);
    Tensor<ADataType> a(HostTensorDescriptor({m, k}, {static_cast<std::size_t>(sa), 1}));
    Tensor<int8_t> q(HostTensorDescriptor({k, n}, {1, static_cast<std::size_t>(sb)}));
    Tensor<ADataType> s(HostTensorDescriptor({k, n}, {0, 1}));
    Tensor<ADataType> c_dev(HostTensorDescriptor({m, n}, {static_cast<std::size_t>(sc), 1})), c_host(c_dev.mDesc);
    const float qlo = Int4 ? -8.f : -128.f, qhi = Int4 ? 7.f : 127.f;
    if(c.init == 1)
    {
        ck::utils::FillUniformDistributionIntegerValue<ADataType>{-1.f, 1.f}(a);
        ck::utils::FillUniformDistributionIntegerValue<ADataType>{1.f, 2.f}(s);
    }
    else
    {
        ck::utils::FillUniformDistribution<ADataType>{-1.f, 1.f}(a);
        ck::utils::FillUniformDistribution<ADataType>{0.004f, 0.06f}(s);
    }
    ck::utils::FillUniformDistributionIntegerValue<int8_t>{qlo, qhi}(q);

    std::vector<uint8_t> packed;
    const void* b_host = q.mData.data();
    std::size_t b_bytes = q.mData.size();
    if(Int4)
    {
        packed.resize(static_cast<std::size_t>(sb) / 2 * n);
        if(wqgemm_pack_int4_host(q.mData.data(), sb, c.K, c.N, packed.data(), sb / 2) != WQ_OK)
        {
            std::cerr << wqgemm_last_error() << std::endl;
            return 1;
        }
        b_host = packed.data(), b_bytes = packed.size();
    }
    DeviceMem a_buf(sizeof(ADataType) * a.mData.size()), b_buf(b_bytes), s_buf(sizeof(ADataType) * s.mData.size()),
        c_buf(sizeof(ADataType) * c_dev.mData.size());
    a_buf.ToDevice(a.mData.data());
    b_buf.ToDevice(b_host);
    s_buf.ToDevice(s.mData.data());

    auto op  = Op{};
    auto arg = op.MakeArgument(static_cast<ADataType*>(a_buf.GetDeviceBuffer()), static_cast<BDataType*>(b_buf.GetDeviceBuffer()),
                               static_cast<ADataType*>(s_buf.GetDeviceBuffer()), static_cast<ADataType*>(c_buf.GetDeviceBuffer()),
                               c.M, c.N, c.K, sa, sb, sc, PassThrough{}, PassThrough{}, PassThrough{});
    arg.prob.k_batch = kbatch;
    DeviceMem workspace(op.GetWorkSpaceSize(&arg));
    op.SetWorkSpacePointer(&arg, workspace.GetDeviceBuffer());
    const std::string name = dev::wq_detail::type_string(arg.prob);
    if(!op.IsSupportedArgument(arg))
    {
        std::cout << name << " does not support this problem" << std::endl;
        return 0;
    }
    const float ms = op.MakeInvoker().Run(arg, stream_config(c.time, n_warmup, n_iter));
    hip_check_error(hipDeviceSynchronize());
    bool pass = true;
    if(c.verify)
    {
        c_buf.FromDevice(c_dev.mData.data());
        if constexpr(std::is_same_v<ADataType, ck::half_t>)
        {
            using Ref = ck::tensor_operation::host::ReferencefpAintBGemm<ADataType, int8_t, ADataType, ADataType, float,
                                                                        PassThrough, PassThrough, PassThrough>;
            auto ref = Ref{};
            ref.MakeInvoker().Run(ref.MakeArgument(a, q, s, c_host, PassThrough{}, PassThrough{}, PassThrough{}));
        }
        else
        {
            // bf16: B = Multiply(q, scale) truncated to bf16, then ReferenceGemm (example/60's verification)
            Tensor<ADataType> b(HostTensorDescriptor({k, n}, {1, k}));
            for(std::size_t j = 0; j < n; ++j)
                for(std::size_t i = 0; i < k; ++i)
                    ck::tensor_operation::element_wise::Multiply{}(b(i, j), q(i, j), s(i, j));
            using Ref = ck::tensor_operation::host::ReferenceGemm<ADataType, ADataType, ADataType, float, PassThrough,
                                                                 PassThrough, PassThrough>;
            auto ref = Ref{};
            ref.MakeInvoker().Run(ref.MakeArgument(a, b, c_host, PassThrough{}, PassThrough{}, PassThrough{}));
        }
        pass = ck::utils::check_err(c_dev, c_host, "Error: Incorrect results!",
                                    std::is_same_v<ADataType, ck::half_t> ? 1e-3 : 1e-2, 1e-3);
        std::cout << (pass ? "verification: PASS" : "verification: FAIL") << std::endl;
    }
    if(c.time)
        report(std::is_same_v<ADataType, ck::half_t> ? "f16" : "bf16", c.M, c.N, c.K, sa, sb, sc, ms,
               sizeof(ADataType) * (m * k + m * n + n) + (Int4 ? k * n / 2 : k * n), name,
               kbatch > 1 ? (", KBatch " + std::to_string(kbatch)).c_str() : "");
    return pass ? 0 : 1;
}

int dispatch_fpaintb(const Common& c, ck::index_t sa, ck::index_t sb, ck::index_t sc, int kbatch, int n_warmup, int n_iter)
{
    switch(c.data_type)
    {
    case 0: return run_fpaintb<ck::half_t, true>(c, sa, sb, sc, kbatch, n_warmup, n_iter);
    case 1: return run_fpaintb<ck::half_t, false>(c, sa, sb, sc, kbatch, n_warmup, n_iter);
    case 2: return run_fpaintb<ck::bhalf_t, true>(c, sa, sb, sc, kbatch, n_warmup, n_iter);
    case 3: return run_fpaintb<ck::bhalf_t, false>(c, sa, sb, sc, kbatch, n_warmup, n_iter);
    default: std::cerr << "data type must be 0..3" << std::endl; return 1;
    }
}

const char* kFpaintbTypes = "arg2: data type (0: f16@int4; 1: f16@int8; 2: bf16@int4; 3: bf16@int8, per-channel scale of A's type)\n";
const char* kLayoutText   = "arg3: matrix layout (1: A[m, k] * B[n, k] = C[m, n])\n"
                            "arg4: verification (0: no; 1: yes)\n"
                            "arg5: initialization (0: no init; 1: integer value; 2: decimal value)\n"
                            "arg6: print tensor value (0: no; 1: yes)\n"
                            "arg7: time kernel (0=no, 1=yes)\n";

int profile_gemm_fpaintb(int argc, char* argv[])
{
    Common c;
    if((argc != 14 && argc != 16) || !parse_common(argc, argv, 14, c))
    {
        std::cout << "arg1: tensor operation (gemm_fpAintB: GEMM with weight-only quantized B)\n"
                  << kFpaintbTypes << kLayoutText << "arg8 to 13: M, N, K, StrideA, StrideB, StrideC\noptional:\n"
                  << "arg14: number of warm-up cycles (default 1)\narg15: number of iterations (default 10)\n";
        return 1;
    }
    auto stride = [](int v, int dflt) { return v < 0 ? dflt : v; };
    const int sa = stride(std::stoi(argv[11]), c.K), sb = stride(std::stoi(argv[12]), c.K), sc = stride(std::stoi(argv[13]), c.N);
    const int n_warmup = argc == 16 ? std::stoi(argv[14]) : 1, n_iter = argc == 16 ? std::stoi(argv[15]) : 10;
    return dispatch_fpaintb(c, sa, sb, sc, 1, n_warmup, n_iter);
}

int profile_gemm_splitk(int argc, char* argv[])
{
    Common c;
    if((argc != 15 && argc != 17) || !parse_common(argc, argv, 15, c))
    {
        std::cout << "arg1: tensor operation (gemm_splitk: Split-K GEMM)\n"
                  << kFpaintbTypes << kLayoutText << "arg8 to 13: M, N, K, StrideA, StrideB, StrideC\n"
                  << "arg14: split k into  mulitiple batch\noptional:\n"
                  << "arg15: number of warm-up cycles (default 1)\narg16: number of iterations (default 10)\n";
        return 1;
    }
    auto stride = [](int v, int dflt) { return v < 0 ? dflt : v; };
    const int sa = stride(std::stoi(argv[11]), c.K), sb = stride(std::stoi(argv[12]), c.K), sc = stride(std::stoi(argv[13]), c.N);
    const int kbatch   = std::stoi(argv[14]);
    const int n_warmup = argc == 17 ? std::stoi(argv[15]) : 1, n_iter = argc == 17 ? std::stoi(argv[16]) : 10;
    return dispatch_fpaintb(c, sa, sb, sc, kbatch, n_warmup, n_iter);
}

// ---- int8 x int8 -> int8 with Activation_Mul_Clamp ---------------------------------------------------------------
int profile_gemm_quantization(int argc, char* argv[])
{
    Common c;
    // under the name "gemm" (profiler/src/profile_gemm.cpp:20-27,51-80) arguments 14, 15 are warm-up cycles and
    // iterations; under "gemm_quantization" argument 14 is the requantisation scale
    const bool as_gemm = std::string(argv[1]) == "gemm";
    if((as_gemm ? (argc != 14 && argc != 16) : (argc != 14 && argc != 15)) || !parse_common(argc, argv, 14, c) || c.data_type != 3)
    {
        if(as_gemm)
        {
            std::cout << "arg1: tensor operation (gemm: GEMM)\narg2: data type (3: int8 -- the requantised int8 GEMM, "
                         "Activation_Mul_Clamp with scale 0.03; the reference's other data types have no quantized operand)\n"
                      << kLayoutText << "arg8 to 13: M, N, K, StrideA, StrideB, StrideC\noptional:\narg14: number of warm-up "
                                        "cycles (default 1)\narg15: number of iterations (default 10)\n";
            return 1;
        }
        std::cout << "arg1: tensor operation (gemm_quantization: int8 GEMM + Activation_Mul_Clamp)\n"
                  << "arg2: data type (3: int8)\n"
                  << kLayoutText << "arg8 to 13: M, N, K, StrideA, StrideB, StrideC\noptional:\narg14: requantisation scale (default 0.03)\n";
        return 1;
    }
    using CDE = ck::tensor_operation::element_wise::Activation_Mul_Clamp<PassThrough>;
    using Op  = dev::DeviceGemmQuantization_gfx950<Row, Col, Row, CDE>;
    auto stride = [](int v, int dflt) { return v < 0 ? dflt : v; };
    const int sa = stride(std::stoi(argv[11]), c.K), sb = stride(std::stoi(argv[12]), c.K), sc = stride(std::stoi(argv[13]), c.N);
    const float requant = (!as_gemm && argc == 15) ? std::stof(argv[14]) : 0.03f;
    const int n_warmup = (as_gemm && argc == 16) ? std::stoi(argv[14]) : 1, n_iter = (as_gemm && argc == 16) ? std::stoi(argv[15]) : 10;
    const auto m = static_cast<std::size_t>(c.M), n = static_cast<std::size_t>(c.N), k = static_cast<std::size_t>(c.K);
    Tensor<int8_t> a(HostTensorDescriptor({m, k}, {static_cast<std::size_t>(sa), 1}));
    Tensor<int8_t> b(HostTensorDescriptor({k, n}, {1, static_cast<std::size_t>(sb)}));
    Tensor<int8_t> e_dev(HostTensorDescriptor({m, n}, {static_cast<std::size_t>(sc), 1})), e_host(e_dev.mDesc);
    const int lim = c.init == 1 ? 5 : 127;
    a.GenerateTensorValue(GeneratorTensor_2<int8_t>{-lim, lim});
    b.GenerateTensorValue(GeneratorTensor_2<int8_t>{-lim, lim});
    DeviceMem a_buf(a.mData.size()), b_buf(b.mData.size()), e_buf(e_dev.mData.size());
    a_buf.ToDevice(a.mData.data());
    b_buf.ToDevice(b.mData.data());
    auto op  = Op{};
    auto cde = CDE{requant, PassThrough{}};
    auto arg = op.MakeArgument(a_buf.GetDeviceBuffer(), b_buf.GetDeviceBuffer(), {}, e_buf.GetDeviceBuffer(), c.M, c.N, c.K,
                               sa, sb, {}, sc, PassThrough{}, PassThrough{}, cde);
    const std::string name = dev::wq_detail::type_string(arg.prob);
    if(!op.IsSupportedArgument(arg))
    {
        std::cout << name << " does not support this problem" << std::endl;
        return 0;
    }
    const float ms = op.MakeInvoker().Run(arg, stream_config(c.time, n_warmup, n_iter));
    hip_check_error(hipDeviceSynchronize());
    bool pass = true;
    if(c.verify)
    {
        e_buf.FromDevice(e_dev.mData.data());
        using Ref = ck::tensor_operation::host::ReferenceGemm<int8_t, int8_t, int8_t, float, PassThrough, PassThrough, CDE>;
        auto ref  = Ref{};
        ref.MakeInvoker().Run(ref.MakeArgument(a, b, e_host, PassThrough{}, PassThrough{}, cde));
        pass = ck::utils::check_err(e_dev, e_host);
        std::cout << (pass ? "verification: PASS" : "verification: FAIL") << std::endl;
    }
    if(c.time)
        report("int8", c.M, c.N, c.K, sa, sb, sc, ms, m * k + k * n + m * n, name);
    return pass ? 0 : 1;
}

// ---- fp8 activations x int4 weights, MultiplyMultiply ----------------------------------------------------------------
template <typename ADataType, bool NoSpecials>
int run_mm(const Common& c, ck::index_t sa, ck::index_t sb, ck::index_t se, int n_warmup, int n_iter, uint64_t rotating)
{
    using CDE = ck::tensor_operation::element_wise::MultiplyMultiply;
    using Op  = dev::DeviceGemmMultiplyMultiply_WeightQuant_gfx950<Row, Col, Row, ADataType, ck::wq_int4x2_t, ck::half_t, CDE,
                                                                  NoSpecials>;
    const auto m = static_cast<std::size_t>(c.M), n = static_cast<std::size_t>(c.N), k = static_cast<std::size_t>(c.K);
    Tensor<ADataType> a(HostTensorDescriptor({m, k}, {static_cast<std::size_t>(sa), 1}));
    Tensor<int8_t> q(HostTensorDescriptor({k, n}, {1, static_cast<std::size_t>(sb)}));
    Tensor<float> d0(HostTensorDescriptor({m, n}, {0, 1})), d1(HostTensorDescriptor({m, n}, {1, 0}));
    Tensor<float> acc(HostTensorDescriptor({m, n}, {n, 1}));
    Tensor<ck::half_t> e_dev(HostTensorDescriptor({m, n}, {static_cast<std::size_t>(se), 1})), e_host(e_dev.mDesc);
    std::mt19937 gen(11939);
    const uint8_t one = std::is_same_v<ADataType, ck::wq_f8_e4m3fn_t> ? 0x38 : 0x40; // 1.0 in E4M3FN / FNUZ
    auto* bytes       = reinterpret_cast<uint8_t*>(a.mData.data());
    for(std::size_t i = 0; i < a.mData.size(); ++i)
    {
        const uint32_t r = gen();
        if(c.init == 1)
        {
            const int v = static_cast<int>(r % 5) - 2; // -2 .. 2
            bytes[i]    = v == 0 ? 0 : static_cast<uint8_t>((v < 0 ? 0x80 : 0) | (one + (std::abs(v) == 2 ? 8 : 0)));
        }
        else
            bytes[i] = static_cast<uint8_t>((r & 0x80) | ((one - 32 + ((r >> 8) % 64)) & 0x7F));
    }
    q.GenerateTensorValue(GeneratorTensor_2<int8_t>{-8, 8});
    if(c.init == 1)
    {
        d0.GenerateTensorValue(GeneratorTensor_2<float>{1, 3});
        d1.GenerateTensorValue(GeneratorTensor_2<float>{1, 3});
    }
    else
    {
        d0.GenerateTensorValue(GeneratorTensor_3<float>{0.0, 1.0});
        d1.GenerateTensorValue(GeneratorTensor_3<float>{0.0, 1.0});
    }
    std::vector<uint8_t> packed(static_cast<std::size_t>(sb) / 2 * n);
    if(wqgemm_pack_int4_host(q.mData.data(), sb, c.K, c.N, packed.data(), sb / 2) != WQ_OK)
    {
        std::cerr << wqgemm_last_error() << std::endl;
        return 1;
    }
    DeviceMem a_buf(a.mData.size()), b_buf(packed.size()), d0_buf(4 * d0.mData.size()), d1_buf(4 * d1.mData.size()),
        e_buf(2 * e_dev.mData.size());
    a_buf.ToDevice(a.mData.data());
    b_buf.ToDevice(packed.data());
    d0_buf.ToDevice(d0.mData.data());
    d1_buf.ToDevice(d1.mData.data());
    auto op  = Op{};
    auto arg = op.MakeArgument(a_buf.GetDeviceBuffer(), b_buf.GetDeviceBuffer(), {d0_buf.GetDeviceBuffer(), d1_buf.GetDeviceBuffer()},
                               e_buf.GetDeviceBuffer(), c.M, c.N, c.K, sa, sb, {0, 0}, se, PassThrough{}, PassThrough{}, CDE{});
    const std::string name = dev::wq_detail::type_string(arg.prob);
    if(!op.IsSupportedArgument(arg))
    {
        std::cout << name << " does not support this problem" << std::endl;
        return 0;
    }
    const int rotating_count =
        rotating_count_for(rotating, a.mData.size() + packed.size() + 4 * (d0.mData.size() + d1.mData.size()), n_iter);
    std::cout << "rotating count: " << rotating_count << std::endl;
    const float ms = op.MakeInvoker().Run(arg, stream_config(c.time, n_warmup, n_iter, rotating_count));
    hip_check_error(hipDeviceSynchronize());
    bool pass = true;
    if(c.verify)
    {
        e_buf.FromDevice(e_dev.mData.data());
        using Ref = ck::tensor_operation::host::ReferenceGemm<ADataType, int8_t, float, float, PassThrough, PassThrough, PassThrough>;
        auto ref  = Ref{};
        ref.MakeInvoker().Run(ref.MakeArgument(a, q, acc, PassThrough{}, PassThrough{}, PassThrough{}));
        for(std::size_t i = 0; i < m; ++i)
            for(std::size_t j = 0; j < n; ++j)
                CDE{}(e_host(i, j), acc(i, j), d0(i, j), d1(i, j));
        pass = ck::utils::check_err(e_dev, e_host);
        std::cout << (pass ? "verification: PASS" : "verification: FAIL") << std::endl;
    }
    if(c.time)
        report("f16", c.M, c.N, c.K, sa, sb, se, ms, m * k + k * n / 2 + 4 * (m + n) + 2 * m * n, name);
    return pass ? 0 : 1;
}

int profile_gemm_multiply_multiply(int argc, char* argv[])
{
    Common c;
    if((argc != 16 && argc != 19) || !parse_common(argc, argv, 16, c) || c.data_type < 8 || c.data_type > 10)
    {
        std::cout << "arg1: tensor operation (gemm_multiply_multiply: GEMM_Multiply_Multiply)\n"
                  << "arg2: data type (8: f8 FNUZ@int4 -> f16, any byte; 9: f8 FNUZ@int4 -> f16, no 0x7F/0xFF/0x80 bytes, fp8 "
                     "matrix core; 10: f8 E4M3FN@int4 -> f16, fp8 matrix core)\n"
                  << kLayoutText << "arg8 to 15: M, N, K, StrideA, StrideB, StrideD0, StrideD1, StrideE\noptional:\n"
                  << "arg16: number of warm-up cycles (default 1)\narg17: number of iterations (default 10)\n"
                  << "arg18: memory for rotating buffer (default 0, size in MB)\n";
        return 1;
    }
    auto stride = [](int v, int dflt) { return v < 0 ? dflt : v; };
    const int sa = stride(std::stoi(argv[11]), c.K), sb = stride(std::stoi(argv[12]), c.K), se = stride(std::stoi(argv[15]), c.N);
    const int n_warmup = argc == 19 ? std::stoi(argv[16]) : 1, n_iter = argc == 19 ? std::stoi(argv[17]) : 10;
    const uint64_t rotating = argc == 19 ? std::stoull(argv[18]) * 1024 * 1024 : 0;
    if(c.data_type == 8)
        return run_mm<ck::f8_t, false>(c, sa, sb, se, n_warmup, n_iter, rotating);
    if(c.data_type == 9)
        return run_mm<ck::f8_t, true>(c, sa, sb, se, n_warmup, n_iter, rotating);
    return run_mm<ck::wq_f8_e4m3fn_t, false>(c, sa, sb, se, n_warmup, n_iter, rotating);
}

// ---- fp8 activations x int4 weights, fp32 scales per 128 x 128 block of both operands ------------------------------------
template <typename ADataType, typename EDataType, bool NoSpecials>
int run_ab_scale(const Common& c, ck::index_t sa, ck::index_t sb, ck::index_t se, int n_warmup, int n_iter, uint64_t rotating)
{
    constexpr ck::index_t BM = 128, BN = 128, BK = 128;
    using Op = dev::DeviceGemmMultiD_ABScale_WeightQuant_gfx950<Row, Col, Row, ADataType, ck::wq_int4x2_t, EDataType, BM, BN, BK, NoSpecials>;
    const auto m = static_cast<std::size_t>(c.M), n = static_cast<std::size_t>(c.N), k = static_cast<std::size_t>(c.K);
    const std::size_t mb = (m + BM - 1) / BM, nb = (n + BN - 1) / BN, kb = (k + BK - 1) / BK;
    Tensor<ADataType> a0(HostTensorDescriptor({m, k}, {static_cast<std::size_t>(sa), 1}));
    Tensor<float> a1(HostTensorDescriptor({mb, kb}, {kb, 1}));
    Tensor<int8_t> q(HostTensorDescriptor({k, n}, {1, static_cast<std::size_t>(sb)}));
    Tensor<float> b1(HostTensorDescriptor({kb, nb}, {1, kb}));
    Tensor<EDataType> e_dev(HostTensorDescriptor({m, n}, {static_cast<std::size_t>(se), 1})), e_host(e_dev.mDesc);
    std::mt19937 gen(11939);
    const uint8_t one = std::is_same_v<ADataType, ck::wq_f8_e4m3fn_t> ? 0x38 : 0x40; // 1.0 in E4M3FN / FNUZ
    auto* bytes       = reinterpret_cast<uint8_t*>(a0.mData.data());
    for(std::size_t i = 0; i < a0.mData.size(); ++i)
    {
        const uint32_t r = gen();
        if(c.init == 1)
        {
            const int v = static_cast<int>(r % 5) - 2; // -2 .. 2, as the reference's GeneratorTensor_2{-2, 2}
            bytes[i]    = v == 0 ? 0 : static_cast<uint8_t>((v < 0 ? 0x80 : 0) | (one + (std::abs(v) == 2 ? 8 : 0)));
        }
        else
            bytes[i] = static_cast<uint8_t>((r & 0x80) | ((one - 32 + ((r >> 8) % 64)) & 0x7F));
    }
    q.GenerateTensorValue(GeneratorTensor_2<int8_t>{-8, 8});
    a1.GenerateTensorValue(GeneratorTensor_3<float>{0, 1.0}); // (profile_gemm_ab_scale_impl.hpp:104-117)
    b1.GenerateTensorValue(GeneratorTensor_3<float>{0, 1.0});
    std::vector<uint8_t> packed(static_cast<std::size_t>(sb) / 2 * n);
    if(wqgemm_pack_int4_host(q.mData.data(), sb, c.K, c.N, packed.data(), sb / 2) != WQ_OK)
    {
        std::cerr << wqgemm_last_error() << std::endl;
        return 1;
    }
    DeviceMem a0_buf(a0.mData.size()), a1_buf(4 * a1.mData.size()), b0_buf(packed.size()), b1_buf(4 * b1.mData.size()),
        e_buf(sizeof(EDataType) * e_dev.mData.size());
    a0_buf.ToDevice(a0.mData.data());
    a1_buf.ToDevice(a1.mData.data());
    b0_buf.ToDevice(packed.data());
    b1_buf.ToDevice(b1.mData.data());
    auto op  = Op{};
    auto arg = op.MakeArgument(a0_buf.GetDeviceBuffer(), b0_buf.GetDeviceBuffer(), {}, e_buf.GetDeviceBuffer(), c.M, c.N, c.K, sa, sb, {},
                               se, a1_buf.GetDeviceBuffer(), b1_buf.GetDeviceBuffer(), PassThrough{}, PassThrough{}, PassThrough{});
    const std::string name = dev::wq_detail::type_string(arg.prob);
    if(!op.IsSupportedArgument(arg))
    {
        std::cout << name << " does not support this problem" << std::endl;
        return 0;
    }
    const int rotating_count = rotating_count_for(rotating, a0.mData.size() + packed.size() + 4 * (a1.mData.size() + b1.mData.size()), n_iter);
    std::cout << "rotating count: " << rotating_count << std::endl;
    const float ms = op.MakeInvoker().Run(arg, stream_config(c.time, n_warmup, n_iter, rotating_count));
    hip_check_error(hipDeviceSynchronize());
    bool pass = true;
    if(c.verify)
    {
        // the reference's host check (profile_gemm_ab_scale_impl.hpp:165-213): both operands scaled to fp32, ReferenceGemm, convert
        e_buf.FromDevice(e_dev.mData.data());
        Tensor<float> a_f(HostTensorDescriptor({m, k}, {k, 1})), b_f(HostTensorDescriptor({k, n}, {1, k})), acc(HostTensorDescriptor({m, n}, {n, 1}));
        for(std::size_t i = 0; i < m; ++i)
            for(std::size_t kk = 0; kk < k; ++kk)
                a_f(i, kk) = ck::type_convert<float>(a0(i, kk)) * a1(i / BM, kk / BK);
        for(std::size_t j = 0; j < n; ++j)
            for(std::size_t kk = 0; kk < k; ++kk)
                b_f(kk, j) = static_cast<float>(q(kk, j)) * b1(kk / BK, j / BN);
        using Ref = ck::tensor_operation::host::ReferenceGemm<float, float, float, float, PassThrough, PassThrough, PassThrough>;
        auto ref  = Ref{};
        ref.MakeInvoker().Run(ref.MakeArgument(a_f, b_f, acc, PassThrough{}, PassThrough{}, PassThrough{}));
        for(std::size_t i = 0; i < m; ++i)
            for(std::size_t j = 0; j < n; ++j)
                e_host(i, j) = ck::type_convert<EDataType>(acc(i, j));
        pass = ck::utils::check_err(e_dev, e_host, "Error: Incorrect results!", 5e-2, 5e-2); // (:263-271, the fp8 tolerances)
        std::cout << (pass ? "verification: PASS" : "verification: FAIL") << std::endl;
    }
    if(c.time)
        report(std::is_same_v<EDataType, ck::bhalf_t> ? "bf16" : "f16", c.M, c.N, c.K, sa, sb, se, ms,
               m * k + k * n / 2 + sizeof(EDataType) * m * n, name);
    return pass ? 0 : 1;
}

// argv positions of profiler/src/profile_gemm_ab_scale.cpp:41-95: the scale-block-tile selector sits at arg4, everything after
// it one position later than in the other operations
int profile_gemm_ab_scale(int argc, char* argv[])
{
    Common c{};
    bool ok = argc == 15 || argc == 18;
    if(ok)
    {
        c.data_type = std::stoi(argv[2]);
        c.layout    = std::stoi(argv[3]);
        c.verify    = std::stoi(argv[5]);
        c.init      = std::stoi(argv[6]);
        c.log       = std::stoi(argv[7]);
        c.time      = std::stoi(argv[8]);
        c.M = std::stoi(argv[9]), c.N = std::stoi(argv[10]), c.K = std::stoi(argv[11]);
        ok = c.layout == 1 && std::stoi(argv[4]) == 0 && c.data_type >= 8 && c.data_type <= 11;
    }
    if(!ok)
    {
        std::cout << "arg1: tensor operation (gemm_ab_scale: GEMM_AB_Scale)\n"
                  << "arg2: data type (8: f8 FNUZ@int4 -> f16, no 0x7F/0xFF/0x80 bytes; 9: the same -> bf16; 10: f8 E4M3FN@int4 -> f16; "
                     "11: -> bf16)\n"
                  << "arg3: matrix layout (1: A[m, k] * B[n, k] = C[m, n])\n"
                  << "arg4: scale block tile (0: ScaleBlockM/N/K = [128, 128, 128])\n"
                  << "arg5: verification (0: no; 1: yes)\narg6: initialization (0: no init; 1: integer value; 2: decimal value)\n"
                  << "arg7: print tensor value (0: no; 1: yes)\narg8: time kernel (0=no, 1=yes)\n"
                  << "arg9 to 14: M, N, K, StrideA, StrideB, StrideE\noptional:\n"
                  << "arg15: number of warm-up cycles (default 1)\narg16: number of iterations (default 10)\n"
                  << "arg17: memory for rotating buffer (default 0, size in MB)\n";
        return 1;
    }
    auto stride = [](int v, int dflt) { return v < 0 ? dflt : v; };
    const int sa = stride(std::stoi(argv[12]), c.K), sb = stride(std::stoi(argv[13]), c.K), se = stride(std::stoi(argv[14]), c.N);
    const int n_warmup = argc == 18 ? std::stoi(argv[15]) : 1, n_iter = argc == 18 ? std::stoi(argv[16]) : 10;
    const uint64_t rotating = argc == 18 ? std::stoull(argv[17]) * 1024 * 1024 : 0;
    switch(c.data_type)
    {
    case 8: return run_ab_scale<ck::f8_t, ck::half_t, true>(c, sa, sb, se, n_warmup, n_iter, rotating);
    case 9: return run_ab_scale<ck::f8_t, ck::bhalf_t, true>(c, sa, sb, se, n_warmup, n_iter, rotating);
    case 10: return run_ab_scale<ck::wq_f8_e4m3fn_t, ck::half_t, false>(c, sa, sb, se, n_warmup, n_iter, rotating);
    default: return run_ab_scale<ck::wq_f8_e4m3fn_t, ck::bhalf_t, false>(c, sa, sb, se, n_warmup, n_iter, rotating);
    }
}

const Register r0("gemm_fpAintB", "GEMM with weight-only quantized B (fp16 / bf16 x int4 / int8)", profile_gemm_fpaintb);
const Register r1("gemm_splitk", "Split-K GEMM with weight-only quantized B", profile_gemm_splitk);
const Register r2("gemm_quantization", "int8 GEMM + Activation_Mul_Clamp", profile_gemm_quantization);
const Register r2g("gemm", "GEMM (data type 3: int8, requantised)", profile_gemm_quantization);
const Register r3("gemm_multiply_multiply", "GEMM_Multiply_Multiply (fp8 x int4)", profile_gemm_multiply_multiply);
const Register r4("gemm_ab_scale", "GEMM_AB_Scale (fp8 x int4, 128 x 128 x 128 block scales of both operands)", profile_gemm_ab_scale);

} // namespace

int main(int argc, char* argv[])
{
    if(argc == 1)
    {
        std::cout << "arg1: tensor operation {\n";
        for(const auto& [name, entry] : registry())
            std::cout << "\t" << name << ": " << entry.first << "\n";
        std::cout << "}" << std::endl;
        return 0;
    }
    const auto found = registry().find(argv[1]);
    if(found == registry().end())
    {
        std::cerr << "cannot find operation: " << argv[1] << std::endl;
        return EXIT_FAILURE;
    }
    return found->second.second(argc, argv);
}
