// fp8 A x packed-int4 B GEMM with per-channel (D0[n]) and per-token (D1[m]) fp32 scales and an optional
// FastGelu, through DeviceGemmMultipleD, in the shape of Composable Kernel's
// example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8.cpp:60-270 (same command line, D layouts
// with stride 0, Perf line, host ReferenceGemm + element-wise pass + check_err).
//
//   example_gemm_multiply_multiply_fp8_int4_gfx950 [verify init time_kernel [M N K [fmt [gelu]]]]
//     init: 1 integer values (exact check), 2 fp8 bytes of 8 binades (tolerance check)
//     fmt:  0 FNUZ bytes, any encoding (fp16 image of A in the workspace + fp16 matrix core), 1 FNUZ bytes on the fp8 matrix core (the generated data
//           holds none of 0x7F / 0xFF / 0x80), 2 E4M3FN bytes on the fp8 matrix core
#include <iostream>
#include <random>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"
#include "ck/library/utility/check_err.hpp"
#include "ck/library/utility/device_memory.hpp"
#include "ck/library/utility/host_tensor.hpp"
#include "ck/library/utility/host_tensor_generator.hpp"
#include "ck/library/reference_tensor_operation/cpu/reference_gemm.hpp"

using Row         = ck::tensor_layout::gemm::RowMajor;
using Col         = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
using MulMul      = ck::tensor_operation::element_wise::MultiplyMultiply;
using MulMulGelu  = ck::tensor_operation::element_wise::MultiplyMultiplyFastGelu;
using EDataType   = ck::half_t;
using BDataType   = ck::wq_int4x2_t;

template <typename ADataType, typename CDEOp, bool NoSpecials>
int run(bool do_verification, int init_method, bool time_kernel, ck::index_t M, ck::index_t N, ck::index_t K)
{
    using DeviceOp = ck::tensor_operation::device::DeviceGemmMultiplyMultiply_WeightQuant_gfx950<
        Row, Col, Row, ADataType, BDataType, EDataType, CDEOp, NoSpecials>;
    using ReferenceGemmInstance =
        ck::tensor_operation::host::ReferenceGemm<ADataType, int8_t, float, float, PassThrough, PassThrough, PassThrough>;

    const auto m = static_cast<std::size_t>(M), n = static_cast<std::size_t>(N), k = static_cast<std::size_t>(K);
    Tensor<ADataType> a_m_k(HostTensorDescriptor({m, k}, {k, 1}));
    Tensor<int8_t> b_k_n(HostTensorDescriptor({k, n}, {1, k}));
    Tensor<float> d0_m_n(HostTensorDescriptor({m, n}, {0, 1})); // per column: row-major, stride 0
    Tensor<float> d1_m_n(HostTensorDescriptor({m, n}, {1, 0})); // per row: column-major, stride 0
    Tensor<EDataType> e_host(HostTensorDescriptor({m, n}, {n, 1}));
    Tensor<EDataType> e_device(HostTensorDescriptor({m, n}, {n, 1}));
    Tensor<float> c_m_n(HostTensorDescriptor({m, n}, {n, 1}));

    std::mt19937 gen(11939);
    // one byte per element; 1.0 is 0x40 in FNUZ (bias 8) and 0x38 in E4M3FN (bias 7)
    constexpr bool ocp   = std::is_same_v<ADataType, ck::wq_f8_e4m3fn_t>;
    const uint8_t one    = ocp ? 0x38 : 0x40;
    auto* a_bytes        = reinterpret_cast<uint8_t*>(a_m_k.mData.data());
    static_assert(sizeof(ADataType) == 1);
    for(std::size_t i = 0; i < a_m_k.mData.size(); ++i)
    {
        const uint32_t r = gen();
        if(init_method == 1) // -2, -1, 0, 1, 2
        {
            const int v = static_cast<int>(r % 5) - 2;
            a_bytes[i]  = v == 0 ? 0 : static_cast<uint8_t>((v < 0 ? 0x80 : 0) | (one + (std::abs(v) == 2 ? 8 : 0)));
        }
        else // sign, exponent field within 4 binades either side of 1.0, any mantissa
            a_bytes[i] = static_cast<uint8_t>((r & 0x80) | ((one - 32 + ((r >> 8) % 64)) & 0x7F));
    }
    b_k_n.GenerateTensorValue(GeneratorTensor_2<int8_t>{-8, 8});
    if(init_method == 1)
    {
        d0_m_n.GenerateTensorValue(GeneratorTensor_2<float>{1, 3});
        d1_m_n.GenerateTensorValue(GeneratorTensor_2<float>{1, 3});
    }
    else
    {
        d0_m_n.GenerateTensorValue(GeneratorTensor_3<float>{0.0, 1.0});
        d1_m_n.GenerateTensorValue(GeneratorTensor_3<float>{0.0, 1.0});
    }

    std::vector<uint8_t> b_packed(k / 2 * n);
    if(wqgemm_pack_int4_host(b_k_n.mData.data(), K, K, N, b_packed.data(), K / 2) != WQ_OK)
    {
        std::cerr << wqgemm_last_error() << std::endl;
        return 1;
    }
    DeviceMem a_buf(a_m_k.mData.size()), b_buf(b_packed.size()), d0_buf(sizeof(float) * d0_m_n.mData.size()),
        d1_buf(sizeof(float) * d1_m_n.mData.size()), e_buf(sizeof(EDataType) * e_device.mData.size());
    a_buf.ToDevice(a_m_k.mData.data());
    b_buf.ToDevice(b_packed.data());
    d0_buf.ToDevice(d0_m_n.mData.data());
    d1_buf.ToDevice(d1_m_n.mData.data());

    auto op       = DeviceOp{};
    auto invoker  = op.MakeInvoker();
    auto argument = op.MakeArgument(a_buf.GetDeviceBuffer(), b_buf.GetDeviceBuffer(),
                                    {d0_buf.GetDeviceBuffer(), d1_buf.GetDeviceBuffer()}, e_buf.GetDeviceBuffer(), M, N,
                                    K, K, K, {0, 0}, N, PassThrough{}, PassThrough{}, CDEOp{});
    if(!op.IsSupportedArgument(argument))
    {
        std::cerr << op.GetTypeString() << " does not support this problem" << std::endl;
        return 0;
    }
    // (FNUZ data of unknown content needs scratch for an fp16 image of A; the argument allocated it.  A caller may
    // instead hand a buffer over, the way the reference's split-K examples do:
    // example/35_splitK_gemm/run_gemm_splitk_reduce_multi_d_example.inc:254-255)
    if(std::is_same_v<ADataType, ck::f8_t> && !NoSpecials && init_method == 1)
    {
        static DeviceMem workspace(op.GetWorkSpaceSize(&argument));
        op.SetWorkSpacePointer(&argument, workspace.GetDeviceBuffer());
    }
    const float ave_time = invoker.Run(argument, StreamConfig{nullptr, time_kernel});
    hip_check_error(hipDeviceSynchronize());
    const std::size_t bytes = m * k + k * n / 2 + sizeof(float) * (m + n) + sizeof(EDataType) * m * n;
    if(time_kernel)
        std::cout << "Perf: " << ave_time << " ms, " << 2.0 * m * n * k / 1.E9 / ave_time << " TFlops, "
                  << bytes / 1.E6 / ave_time << " GB/s, " << ck::tensor_operation::device::wq_detail::type_string(argument.prob)
                  << std::endl;
    if(!do_verification)
        return 0;

    e_buf.FromDevice(e_device.mData.data());
    auto ref     = ReferenceGemmInstance{};
    auto ref_inv = ref.MakeInvoker();
    ref_inv.Run(ref.MakeArgument(a_m_k, b_k_n, c_m_n, PassThrough{}, PassThrough{}, PassThrough{}));
    const CDEOp cde{};
    for(std::size_t i = 0; i < m; ++i)
        for(std::size_t j = 0; j < n; ++j)
            cde(e_host(i, j), c_m_n(i, j), d0_m_n(i, j), d1_m_n(i, j));
    const bool ok = ck::utils::check_err(e_device, e_host);
    std::cout << (ok ? "verification: PASS" : "verification: FAIL") << std::endl;
    return ok ? 0 : 1;
}

int main(int argc, char* argv[])
{
    bool do_verification = true, time_kernel = false;
    int init_method = 1, fmt = 1, gelu = 0;
    ck::index_t M = 1024, N = 1024, K = 1024;
    if(argc >= 4)
    {
        do_verification = std::stoi(argv[1]);
        init_method     = std::stoi(argv[2]);
        time_kernel     = std::stoi(argv[3]);
    }
    if(argc >= 7)
        M = std::stoi(argv[4]), N = std::stoi(argv[5]), K = std::stoi(argv[6]);
    if(argc >= 8)
        fmt = std::stoi(argv[7]);
    if(argc >= 9)
        gelu = std::stoi(argv[8]);
    if(argc == 2 || argc == 3 || argc == 5 || argc == 6 || argc > 9)
    {
        std::cerr << "arg1: verification (0=no, 1=yes)\narg2: initialization (1=integer value, 2=fp8 bytes)\n"
                     "arg3: time kernel (0=no, 1=yes)\narg4 to 6: M, N, K\narg7: 0 FNUZ via fp16, 1 FNUZ matrix core, "
                     "2 E4M3FN matrix core\narg8: 1 = FastGelu after the scales\n";
        return 1;
    }
    if(fmt == 2)
        return gelu ? run<ck::wq_f8_e4m3fn_t, MulMulGelu, false>(do_verification, init_method, time_kernel, M, N, K)
                    : run<ck::wq_f8_e4m3fn_t, MulMul, false>(do_verification, init_method, time_kernel, M, N, K);
    if(fmt == 1)
        return gelu ? run<ck::f8_t, MulMulGelu, true>(do_verification, init_method, time_kernel, M, N, K)
                    : run<ck::f8_t, MulMul, true>(do_verification, init_method, time_kernel, M, N, K);
    return gelu ? run<ck::f8_t, MulMulGelu, false>(do_verification, init_method, time_kernel, M, N, K)
                : run<ck::f8_t, MulMul, false>(do_verification, init_method, time_kernel, M, N, K);
}
