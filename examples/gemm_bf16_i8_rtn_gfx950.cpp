// bf16 A x int8 B GEMM with a per-column bf16 weight scale (Multiply on B) whose fp32 accumulator leaves as bf16 rounded
// to nearest even: the C element-wise operation ConvertBF16RTN of Composable Kernel's
// example/01_gemm/gemm_xdl_bf16_rtn.cpp:9-39 on the weight-quantized operands of
// example/60_gemm_multi_ABD/gemm_multi_ABD_xdl_fastgelu_bf16_i8.cpp:60-272 (same command line shape, Perf line, host
// ReferenceGemm with the functor as its C operation, check_err).
//
//   example_gemm_bf16_i8_rtn_gfx950 [verify init time_kernel [M N K]]
//     init: 1 half-integer activations, integer scales (sums exact in fp32 in any order: bit-exact check, with ties),
//           2 decimal values (check_err with rtol 1e-2, atol 1e-3: one bf16 step)
#include <iostream>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"
#include "ck/library/utility/check_err.hpp"
#include "ck/library/utility/device_memory.hpp"
#include "ck/library/utility/host_tensor.hpp"
#include "ck/library/utility/host_tensor_generator.hpp"
#include "ck/library/reference_tensor_operation/cpu/reference_gemm.hpp"

using Row            = ck::tensor_layout::gemm::RowMajor;
using Col            = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough    = ck::tensor_operation::element_wise::PassThrough;
using Multiply       = ck::tensor_operation::element_wise::Multiply;
using ConvertBF16RTN = ck::tensor_operation::element_wise::ConvertBF16RTN;
using BF16           = ck::bhalf_t;

using DeviceOp = ck::tensor_operation::device::DeviceGemmMultiABD_gfx950<Row, Col, Row, BF16, int8_t, BF16, true, ConvertBF16RTN>;
using ReferenceGemmInstance =
    ck::tensor_operation::host::ReferenceGemm<BF16, BF16, BF16, float, PassThrough, PassThrough, ConvertBF16RTN>;

int main(int argc, char* argv[])
{
    bool do_verification = true, time_kernel = false;
    int init_method = 1;
    ck::index_t M = 512, N = 768, K = 1024;
    if(argc >= 4)
    {
        do_verification = std::stoi(argv[1]);
        init_method     = std::stoi(argv[2]);
        time_kernel     = std::stoi(argv[3]);
    }
    if(argc >= 7)
    {
        M = std::stoi(argv[4]);
        N = std::stoi(argv[5]);
        K = std::stoi(argv[6]);
    }
    const auto m = static_cast<std::size_t>(M), n = static_cast<std::size_t>(N), k = static_cast<std::size_t>(K);
    Tensor<BF16> a_m_k(HostTensorDescriptor({m, k}, {k, 1}));
    Tensor<int8_t> b0_k_n(HostTensorDescriptor({k, n}, {1, k}));
    Tensor<BF16> b1_k_n(HostTensorDescriptor({k, n}, {0, 1})); // one scale per column
    Tensor<BF16> b_k_n(HostTensorDescriptor({k, n}, {1, k}));
    Tensor<BF16> e_host(HostTensorDescriptor({m, n}, {n, 1})), e_device(HostTensorDescriptor({m, n}, {n, 1}));

    if(init_method == 1)
    {
        a_m_k.GenerateTensorValue(GeneratorTensor_2<BF16>{-4, 5});
        for(auto& v : a_m_k.mData) // halves: sums of products reach bit patterns that sit exactly between two bf16 values
            v = ck::type_convert<BF16>(ck::type_convert<float>(v) * 0.5f);
        b0_k_n.GenerateTensorValue(GeneratorTensor_2<int8_t>{-16, 16});
        b1_k_n.GenerateTensorValue(GeneratorTensor_2<BF16>{1, 4});
    }
    else
    {
        a_m_k.GenerateTensorValue(GeneratorTensor_3<BF16>{-0.5, 0.5});
        b0_k_n.GenerateTensorValue(GeneratorTensor_2<int8_t>{-128, 128});
        b1_k_n.GenerateTensorValue(GeneratorTensor_3<BF16>{0.004, 0.06});
    }

    DeviceMem a_buf(sizeof(BF16) * a_m_k.mData.size()), b0_buf(b0_k_n.mData.size()), b1_buf(sizeof(BF16) * b1_k_n.mData.size()),
        e_buf(sizeof(BF16) * e_device.mData.size());
    a_buf.ToDevice(a_m_k.mData.data());
    b0_buf.ToDevice(b0_k_n.mData.data());
    b1_buf.ToDevice(b1_k_n.mData.data());

    auto op       = DeviceOp{};
    auto invoker  = op.MakeInvoker();
    auto argument = op.MakeArgument({a_buf.GetDeviceBuffer()}, {b0_buf.GetDeviceBuffer(), b1_buf.GetDeviceBuffer()}, {},
                                    e_buf.GetDeviceBuffer(), M, N, K, {K}, {K, 0}, {}, N, PassThrough{}, Multiply{},
                                    ConvertBF16RTN{});
    if(!op.IsSupportedArgument(argument))
    {
        std::cerr << op.GetTypeString() << " does not support this problem" << std::endl;
        return 0;
    }
    const float ave_time = invoker.Run(argument, StreamConfig{nullptr, time_kernel});
    hip_check_error(hipDeviceSynchronize());
    if(time_kernel)
        std::cout << "Perf: " << ave_time << " ms, " << 2.0 * m * n * k / 1.E9 / ave_time << " TFlops, "
                  << (sizeof(BF16) * (m * k + n + m * n) + k * n) / 1.E6 / ave_time << " GB/s, "
                  << ck::tensor_operation::device::wq_detail::type_string(argument.prob) << std::endl;
    if(!do_verification)
        return 0;

    e_buf.FromDevice(e_device.mData.data());
    for(std::size_t j = 0; j < n; ++j)
        for(std::size_t i = 0; i < k; ++i)
            Multiply{}(b_k_n(i, j), b0_k_n(i, j), b1_k_n(i, j));
    auto ref     = ReferenceGemmInstance{};
    auto ref_inv = ref.MakeInvoker();
    ref_inv.Run(ref.MakeArgument(a_m_k, b_k_n, e_host, PassThrough{}, PassThrough{}, ConvertBF16RTN{}));
    const bool ok = init_method == 1 ? ck::utils::check_err(e_device, e_host, "Error: Incorrect results!", 0, 0)
                                     // one bf16 step (2^-8 relative) where fp32 summation order moves a sum across a
                                     // rounding boundary
                                     : ck::utils::check_err(e_device, e_host, "Error: Incorrect results!", 1e-2, 1e-3);
    std::cout << (ok ? "verification: PASS" : "verification: FAIL") << std::endl;
    return ok ? 0 : 1;
}
