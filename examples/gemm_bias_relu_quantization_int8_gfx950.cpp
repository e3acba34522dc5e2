// int8 x int8 GEMM + int32 per-column bias + Relu + requantisation (Add_Activation_Mul_Clamp<Relu>) through
// DeviceGemmMultipleD with one D tensor, in the shape of Composable Kernel's
// example/14_gemm_quantization/gemm_xdl_bias_relu_quantization_int8.cpp:105-235 (fixed 1024^3 problem, bias of N
// int32 values passed with stride 0, GeneratorTensor_2{-128,127} inputs, requant scale 0.03, int32 host
// reference followed by the functor, exact comparison).
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

using Row          = ck::tensor_layout::gemm::RowMajor;
using Col          = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough  = ck::tensor_operation::element_wise::PassThrough;
using ActivationOp = ck::tensor_operation::element_wise::Relu;
using CDEElementOp = ck::tensor_operation::element_wise::Add_Activation_Mul_Clamp<ActivationOp>;

using DeviceGemmInstance = ck::tensor_operation::device::DeviceGemmBiasQuantization_gfx950<Row, Col, Row, ActivationOp>;
using ReferenceGemmInstance =
    ck::tensor_operation::host::ReferenceGemm<int8_t, int8_t, int32_t, int32_t, PassThrough, PassThrough, PassThrough>;

int main(int argc, char* argv[])
{
    const bool time_kernel = argc > 1 && std::stoi(argv[1]) != 0;
    const ck::index_t M = argc > 2 ? std::stoi(argv[2]) : 1024, N = M, K = M;
    const float requant_scale = 0.03f;
    const auto m = static_cast<std::size_t>(M), n = static_cast<std::size_t>(N), k = static_cast<std::size_t>(K);

    Tensor<int8_t> a_m_k(HostTensorDescriptor({m, k}, {k, 1}));
    Tensor<int8_t> b_k_n(HostTensorDescriptor({k, n}, {1, k}));
    Tensor<int32_t> bias_n(HostTensorDescriptor({n}, {1}));
    Tensor<int8_t> e_host(HostTensorDescriptor({m, n}, {n, 1}));
    Tensor<int8_t> e_device(HostTensorDescriptor({m, n}, {n, 1}));
    a_m_k.GenerateTensorValue(GeneratorTensor_2<int8_t>{-128, 127});
    b_k_n.GenerateTensorValue(GeneratorTensor_2<int8_t>{-128, 127});
    bias_n.GenerateTensorValue(GeneratorTensor_2<int32_t>{-128, 127});

    DeviceMem a_buf(a_m_k.mData.size()), b_buf(b_k_n.mData.size()), bias_buf(sizeof(int32_t) * bias_n.mData.size()),
        e_buf(e_device.mData.size());
    a_buf.ToDevice(a_m_k.mData.data());
    b_buf.ToDevice(b_k_n.mData.data());
    bias_buf.ToDevice(bias_n.mData.data());

    auto gemm     = DeviceGemmInstance{};
    auto invoker  = gemm.MakeInvoker();
    auto cde_op   = CDEElementOp{requant_scale, ActivationOp{}};
    auto argument = gemm.MakeArgument(a_buf.GetDeviceBuffer(), b_buf.GetDeviceBuffer(), {bias_buf.GetDeviceBuffer()},
                                      e_buf.GetDeviceBuffer(), M, N, K, K, K, {0}, N, PassThrough{}, PassThrough{},
                                      cde_op);
    if(!gemm.IsSupportedArgument(argument))
        throw std::runtime_error("wrong! device_gemm does not support this GEMM problem");
    const float ave_time = invoker.Run(argument, StreamConfig{nullptr, time_kernel});
    hip_check_error(hipDeviceSynchronize());
    if(time_kernel)
        std::cout << "Perf: " << ave_time << " ms, " << 2.0 * m * n * k / 1.E9 / ave_time << " TFlops, "
                  << (m * k + k * n + m * n) / 1.E6 / ave_time << " GB/s, " << gemm.GetTypeString() << std::endl;
    e_buf.FromDevice(e_device.mData.data());

    Tensor<int32_t> c_m_n(HostTensorDescriptor({m, n}, {n, 1}));
    auto ref     = ReferenceGemmInstance{};
    auto ref_inv = ref.MakeInvoker();
    ref_inv.Run(ref.MakeArgument(a_m_k, b_k_n, c_m_n, PassThrough{}, PassThrough{}, PassThrough{}));
    for(std::size_t i = 0; i < m; ++i)
        for(std::size_t j = 0; j < n; ++j)
            cde_op(e_host(i, j), c_m_n(i, j), bias_n(j));
    const bool ok = ck::utils::check_err(e_device, e_host);
    std::cout << (ok ? "verification: PASS" : "verification: FAIL") << std::endl;
    return ok ? 0 : 1;
}
