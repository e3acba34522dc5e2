// fp8 A x packed-int4 B GEMM with fp32 scales per (128 x 128) block of BOTH operands, through
// DeviceGemmMultipleD_ABScale, in the shape of Composable Kernel's
// example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:84-330 (same command line, same scale
// tensors and strides, Perf line, host check: operands scaled to fp32, ReferenceGemm<float, float>, convert, check_err
// at 5e-2) -- with weight-quantized B in place of fp8 B.
//
//   example_gemm_ab_scale_fp8_int4_gfx950 [verify init time_kernel [M N K [fmt [per_token]]]]
//     init: 1 integer values and power-of-two scales (every block sum exact: bit-exact check), otherwise fp8 bytes of
//           8 binades and uniform scales in (0, 1)
//     fmt:  1 FNUZ bytes on the fp8 matrix core (the generated data holds none of 0x7F / 0xFF / 0x80), 2 E4M3FN bytes
//     per_token: 1 = one activation scale per row and 128 k (ScaleBlockM = 1) instead of per 128 rows
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
using EDataType   = ck::bhalf_t;
using BDataType   = ck::wq_int4x2_t;

static constexpr ck::index_t Scale_Block_N = 128;
static constexpr ck::index_t Scale_Block_K = 128;

template <typename ADataType, bool NoSpecials, ck::index_t Scale_Block_M>
int run(bool do_verification, int init_method, bool time_kernel, ck::index_t M, ck::index_t N, ck::index_t K)
{
    using DeviceOp = ck::tensor_operation::device::DeviceGemmMultiD_ABScale_WeightQuant_gfx950<
        Row, Col, Row, ADataType, BDataType, EDataType, Scale_Block_M, Scale_Block_N, Scale_Block_K, NoSpecials>;
    using ReferenceGemmInstance =
        ck::tensor_operation::host::ReferenceGemm<float, float, float, float, PassThrough, PassThrough, PassThrough>;

    const auto m = static_cast<std::size_t>(M), n = static_cast<std::size_t>(N), k = static_cast<std::size_t>(K);
    const std::size_t mb = (m + Scale_Block_M - 1) / Scale_Block_M, nb = (n + Scale_Block_N - 1) / Scale_Block_N,
                      kb = (k + Scale_Block_K - 1) / Scale_Block_K;
    Tensor<ADataType> a0_m_k(HostTensorDescriptor({m, k}, {k, 1}));
    Tensor<float> a1_m_k(HostTensorDescriptor({mb, kb}, {kb, 1}));
    Tensor<int8_t> b0_k_n(HostTensorDescriptor({k, n}, {1, k}));
    Tensor<float> b1_k_n(HostTensorDescriptor({kb, nb}, {1, kb}));
    Tensor<EDataType> e_host(HostTensorDescriptor({m, n}, {n, 1}));
    Tensor<EDataType> e_device(HostTensorDescriptor({m, n}, {n, 1}));
    std::cout << "a0_m_k: " << a0_m_k.mDesc << "\na1_m_k: " << a1_m_k.mDesc << "\nb0_k_n: " << b0_k_n.mDesc
              << "\nb1_k_n: " << b1_k_n.mDesc << "\ne_m_n: " << e_host.mDesc << std::endl;

    std::mt19937 gen(11939);
    constexpr bool ocp = std::is_same_v<ADataType, ck::wq_f8_e4m3fn_t>;
    const uint8_t one  = ocp ? 0x38 : 0x40; // 1.0: bias 7 / bias 8
    auto* a_bytes      = reinterpret_cast<uint8_t*>(a0_m_k.mData.data());
    static_assert(sizeof(ADataType) == 1);
    for(std::size_t i = 0; i < a0_m_k.mData.size(); ++i)
    {
        const uint32_t r = gen();
        if(init_method == 1) // -2, -1, 0, 1, 2
        {
            const int v = static_cast<int>(r % 5) - 2;
            a_bytes[i]  = v == 0 ? 0 : static_cast<uint8_t>((v < 0 ? 0x80 : 0) | (one + (std::abs(v) == 2 ? 8 : 0)));
        }
        else
            a_bytes[i] = static_cast<uint8_t>((r & 0x80) | ((one - 32 + ((r >> 8) % 64)) & 0x7F));
    }
    b0_k_n.GenerateTensorValue(GeneratorTensor_2<int8_t>{-8, 8});
    if(init_method == 1)
    {
        for(auto& v : a1_m_k.mData)
            v = std::ldexp(1.0f, static_cast<int>(gen() % 4) - 2);
        for(auto& v : b1_k_n.mData)
            v = std::ldexp(1.0f, static_cast<int>(gen() % 4) - 2);
    }
    else
    {
        a1_m_k.GenerateTensorValue(GeneratorTensor_3<float>{0, 1.0});
        b1_k_n.GenerateTensorValue(GeneratorTensor_3<float>{0, 1.0});
    }

    std::vector<uint8_t> b_packed(k / 2 * n);
    if(wqgemm_pack_int4_host(b0_k_n.mData.data(), K, K, N, b_packed.data(), K / 2) != WQ_OK)
    {
        std::cerr << wqgemm_last_error() << std::endl;
        return 1;
    }
    DeviceMem a0_buf(a0_m_k.mData.size()), a1_buf(sizeof(float) * a1_m_k.mData.size()), b0_buf(b_packed.size()),
        b1_buf(sizeof(float) * b1_k_n.mData.size()), e_buf(sizeof(EDataType) * e_device.mData.size());
    a0_buf.ToDevice(a0_m_k.mData.data());
    a1_buf.ToDevice(a1_m_k.mData.data());
    b0_buf.ToDevice(b_packed.data());
    b1_buf.ToDevice(b1_k_n.mData.data());

    auto op       = DeviceOp{};
    auto invoker  = op.MakeInvoker();
    auto argument = op.MakeArgument(a0_buf.GetDeviceBuffer(), b0_buf.GetDeviceBuffer(), {}, e_buf.GetDeviceBuffer(), M, N, K, K, K,
                                    {}, N, a1_buf.GetDeviceBuffer(), b1_buf.GetDeviceBuffer(), PassThrough{}, PassThrough{},
                                    PassThrough{});
    if(!op.IsSupportedArgument(argument))
        throw std::runtime_error("wrong! device_gemm with the specified compilation parameters does not support this GEMM problem");
    const float ave_time = invoker.Run(argument, StreamConfig{nullptr, time_kernel, 20, 50});
    hip_check_error(hipDeviceSynchronize());
    const std::size_t bytes = m * k + k * n / 2 + sizeof(EDataType) * m * n;
    if(time_kernel)
        std::cout << "Perf: " << ave_time << " ms, " << 2.0 * m * n * k / 1.E9 / ave_time << " TFlops, "
                  << bytes / 1.E6 / ave_time << " GB/s, " << ck::tensor_operation::device::wq_detail::type_string(argument.prob)
                  << std::endl;
    if(!do_verification)
        return 0;

    e_buf.FromDevice(e_device.mData.data());
    Tensor<float> c_m_n(HostTensorDescriptor({m, n}, {n, 1}));
    Tensor<float> a_m_k(HostTensorDescriptor({m, k}, {k, 1}));
    Tensor<float> b_k_n(HostTensorDescriptor({k, n}, {1, k}));
    for(std::size_t i = 0; i < m; ++i)
        for(std::size_t kk = 0; kk < k; ++kk)
            a_m_k(i, kk) = ck::type_convert<float>(a0_m_k(i, kk)) * a1_m_k(i / Scale_Block_M, kk / Scale_Block_K);
    for(std::size_t j = 0; j < n; ++j)
        for(std::size_t kk = 0; kk < k; ++kk)
            b_k_n(kk, j) = static_cast<float>(b0_k_n(kk, j)) * b1_k_n(kk / Scale_Block_K, j / Scale_Block_N);
    auto ref     = ReferenceGemmInstance{};
    auto ref_inv = ref.MakeInvoker();
    ref_inv.Run(ref.MakeArgument(a_m_k, b_k_n, c_m_n, PassThrough{}, PassThrough{}, PassThrough{}));
    for(std::size_t i = 0; i < m; ++i)
        for(std::size_t j = 0; j < n; ++j)
            e_host(i, j) = ck::type_convert<EDataType>(c_m_n(i, j));
    const bool ok = init_method == 1 ? ck::utils::check_err(e_device, e_host, "Error: Incorrect results!", 0, 0)
                                     : ck::utils::check_err(e_device, e_host, "Error: Incorrect results!", 5e-2, 5e-2);
    std::cout << (ok ? "verification: PASS" : "verification: FAIL") << std::endl;
    return ok ? 0 : 1;
}

int main(int argc, char* argv[])
{
    bool do_verification = true, time_kernel = false;
    int init_method = 1, fmt = 2;
    ck::index_t M = 3840, N = 4096, K = 4096;
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
    const bool per_token = argc >= 9 && std::stoi(argv[8]) != 0;
    if(argc == 2 || argc == 3 || argc == 5 || argc == 6 || argc > 9)
    {
        std::cerr << "arg1: verification (0=no, 1=yes)\narg2: initialization (1=integer value, 2=decimal value)\n"
                     "arg3: time kernel (0=no, 1=yes)\narg4 to 6: M, N, K\narg7: 1 FNUZ bytes, 2 E4M3FN bytes\n"
                     "arg8: 1 = one activation scale per token (ScaleBlockM = 1)\n";
        return 1;
    }
    if(per_token)
        return fmt == 2 ? run<ck::wq_f8_e4m3fn_t, false, 1>(do_verification, init_method, time_kernel, M, N, K)
                        : run<ck::f8_t, true, 1>(do_verification, init_method, time_kernel, M, N, K);
    return fmt == 2 ? run<ck::wq_f8_e4m3fn_t, false, 128>(do_verification, init_method, time_kernel, M, N, K)
                    : run<ck::f8_t, true, 128>(do_verification, init_method, time_kernel, M, N, K);
}
