// fp16 A x packed-int4 B GEMM with a per-channel fp16 scale on MI355X, driven through the CK-style
// client API (DeviceGemm_dequantB).  Same command line, input generation, Perf line and
// verification flow as Composable Kernel's example/64_fpAintB_gemm (run_gemm_example.inc:6-172,
// common.hpp:85-123), with the weights drawn from the int4 range and packed two per byte.
//
//   example_fpAintB_gemm_gfx950 [verify init time_kernel [M N K StrideA StrideB StrideC [preshuffle]]]
//     init: 0 none, 1 integer values, 2 decimal values
//     preshuffle: 1 = store the packed weights in the engine's pre-shuffled order first (same results, faster fetches)
#include <iostream>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_fpAintB_gemm_mfma_gfx950.hpp"
#include "ck/library/utility/check_err.hpp"
#include "ck/library/utility/device_memory.hpp"
#include "ck/library/utility/fill.hpp"
#include "ck/library/utility/host_tensor.hpp"
#include "ck/library/reference_tensor_operation/cpu/reference_fpAintB_gemm.hpp"

using Row         = ck::tensor_layout::gemm::RowMajor;
using Col         = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;

using ADataType     = ck::half_t;
using QuantDataType = int8_t;          // host-side weights, values -8..7
using ScaleDataType = ck::half_t;
using CDataType     = ck::half_t;

using DeviceGemmInstance = // device-side weights: packed two per byte
    ck::tensor_operation::device::DeviceFpAintBGemm_Mfma_gfx950<Row, Col, Row, ADataType, ck::wq_int4x2_t, ScaleDataType,
                                                                CDataType, PassThrough, PassThrough, PassThrough>;
using DeviceGemmInstancePreshuffled =
    ck::tensor_operation::device::DeviceFpAintBGemm_Mfma_gfx950<Row, Col, Row, ADataType, ck::wq_int4x2_preshuffled_t,
                                                                ScaleDataType, CDataType, PassThrough, PassThrough, PassThrough>;
using ReferenceGemmInstance =
    ck::tensor_operation::host::ReferencefpAintBGemm<ADataType, QuantDataType, ScaleDataType, CDataType, float,
                                                     PassThrough, PassThrough, PassThrough>;

template <typename DeviceGemmInstance, bool Preshuffle>
int run(bool do_verification, int init_method, bool time_kernel, ck::index_t M, ck::index_t N, ck::index_t K, ck::index_t StrideA,
        ck::index_t StrideB, ck::index_t StrideC);

int main(int argc, char* argv[])
{
    bool do_verification = true;
    int init_method      = 1;
    bool time_kernel     = false;
    bool preshuffle      = false;
    ck::index_t M = 3840, N = 4096, K = 4096, StrideA = 4096, StrideB = 4096, StrideC = 4096;
    if(argc == 4 || argc == 10 || argc == 11)
    {
        do_verification = std::stoi(argv[1]);
        init_method     = std::stoi(argv[2]);
        time_kernel     = std::stoi(argv[3]);
        if(argc == 11)
            preshuffle = std::stoi(argv[10]) != 0;
        if(argc >= 10)
        {
            M = std::stoi(argv[4]), N = std::stoi(argv[5]), K = std::stoi(argv[6]);
            StrideA = std::stoi(argv[7]), StrideB = std::stoi(argv[8]), StrideC = std::stoi(argv[9]);
        }
    }
    else if(argc != 1)
    {
        std::cerr << "arg1: verification (0=no, 1=yes)\narg2: initialization (0=no init, 1=integer value, "
                     "2=decimal value)\narg3: time kernel (0=no, 1=yes)\narg4 to 9: M, N, K, StrideA, StrideB, StrideC\n"
                     "arg10: 1 = pre-shuffled weight order\n";
        return 1;
    }
    return preshuffle ? run<DeviceGemmInstancePreshuffled, true>(do_verification, init_method, time_kernel, M, N, K, StrideA, StrideB, StrideC)
                      : run<DeviceGemmInstance, false>(do_verification, init_method, time_kernel, M, N, K, StrideA, StrideB, StrideC);
}

template <typename DeviceGemmInstance, bool Preshuffle>
int run(bool do_verification, int init_method, bool time_kernel, ck::index_t M, ck::index_t N, ck::index_t K, ck::index_t StrideA,
        ck::index_t StrideB, ck::index_t StrideC)
{
    using BDataType = std::conditional_t<Preshuffle, ck::wq_int4x2_preshuffled_t, ck::wq_int4x2_t>;
    const auto m = static_cast<std::size_t>(M), n = static_cast<std::size_t>(N), k = static_cast<std::size_t>(K);
    Tensor<ADataType> a_m_k(HostTensorDescriptor({m, k}, {static_cast<std::size_t>(StrideA), 1}));
    Tensor<QuantDataType> quant_b_k_n(HostTensorDescriptor({k, n}, {1, static_cast<std::size_t>(StrideB)}));
    Tensor<ScaleDataType> scale_k_n(HostTensorDescriptor({k, n}, {0, 1})); // [1, N] broadcast over k
    Tensor<CDataType> c_host(HostTensorDescriptor({m, n}, {static_cast<std::size_t>(StrideC), 1}));
    Tensor<CDataType> c_device(HostTensorDescriptor({m, n}, {static_cast<std::size_t>(StrideC), 1}));

    if(init_method == 1)
    {
        ck::utils::FillUniformDistributionIntegerValue<ADataType>{-1.f, 1.f}(a_m_k);
        ck::utils::FillUniformDistributionIntegerValue<QuantDataType>{-8.f, 7.f}(quant_b_k_n);
        ck::utils::FillUniformDistributionIntegerValue<ScaleDataType>{-1.f, 1.f}(scale_k_n);
    }
    else if(init_method != 0)
    {
        ck::utils::FillUniformDistribution<ADataType>{-1.f, 1.f}(a_m_k);
        ck::utils::FillUniformDistributionIntegerValue<QuantDataType>{-8.f, 7.f}(quant_b_k_n);
        ck::utils::FillUniformDistribution<ScaleDataType>{-1.f, 1.f}(scale_k_n);
    }
    std::cout << "a_m_k: " << a_m_k.mDesc << "\nb_k_n: " << quant_b_k_n.mDesc << "\nscale_k_n: " << scale_k_n.mDesc
              << "\nc_m_n: " << c_host.mDesc << std::endl;

    // pack the weights (the int4 counterpart of example/64's UnsignedWeightPreprocessor)
    const std::size_t packed_ld = static_cast<std::size_t>(StrideB) / 2;
    std::vector<uint8_t> b_packed(packed_ld * n);
    if(wqgemm_pack_int4_host(quant_b_k_n.mData.data(), StrideB, K, N, b_packed.data(), static_cast<int>(packed_ld)) != WQ_OK)
    {
        std::cerr << wqgemm_last_error() << std::endl;
        return 1;
    }
    if constexpr(Preshuffle)
    {
        // offline step of a deployment: the same bytes in the order the kernels fetch them (K pitch = StrideB)
        std::vector<uint8_t> pre(b_packed.size());
        if(wqgemm_preshuffle_int4_host(b_packed.data(), static_cast<int>(packed_ld), K, N, pre.data(), StrideB) != WQ_OK)
        {
            std::cerr << wqgemm_last_error() << std::endl;
            return 1;
        }
        b_packed.swap(pre);
    }

    DeviceMem a_buf(sizeof(ADataType) * a_m_k.mDesc.GetElementSpaceSize());
    DeviceMem b_buf(b_packed.size());
    DeviceMem s_buf(sizeof(ScaleDataType) * scale_k_n.mDesc.GetElementSpaceSize());
    DeviceMem c_buf(sizeof(CDataType) * c_device.mDesc.GetElementSpaceSize());
    a_buf.ToDevice(a_m_k.mData.data());
    b_buf.ToDevice(b_packed.data());
    s_buf.ToDevice(scale_k_n.mData.data());

    auto gemm     = DeviceGemmInstance{};
    auto invoker  = gemm.MakeInvoker();
    auto argument = gemm.MakeArgument(static_cast<ADataType*>(a_buf.GetDeviceBuffer()),
                                      static_cast<BDataType*>(b_buf.GetDeviceBuffer()),
                                      static_cast<ScaleDataType*>(s_buf.GetDeviceBuffer()),
                                      static_cast<CDataType*>(c_buf.GetDeviceBuffer()), M, N, K, StrideA, StrideB,
                                      StrideC, PassThrough{}, PassThrough{}, PassThrough{});
    if(!gemm.IsSupportedArgument(argument))
    {
        std::cerr << gemm.GetTypeString() << " does not support this problem" << std::endl;
        return 0;
    }
    const float ave_time = invoker.Run(argument, StreamConfig{nullptr, time_kernel});
    hip_check_error(hipDeviceSynchronize());

    const std::size_t flop = 2 * m * n * k;
    // the reference's byte formula (one byte per B element) next to the packed-int4 one
    const std::size_t bytes_ck   = sizeof(ADataType) * m * k + 1 * k * n + sizeof(CDataType) * m * n;
    const std::size_t bytes_real = sizeof(ADataType) * m * k + k * n / 2 + sizeof(ScaleDataType) * n + sizeof(CDataType) * m * n;
    if(time_kernel)
        std::cout << "Perf: " << ave_time << " ms, " << static_cast<float>(flop) / 1.E9 / ave_time << " TFlops, "
                  << bytes_ck / 1.E6 / ave_time << " GB/s (packed int4: " << bytes_real / 1.E6 / ave_time << " GB/s), "
                  << gemm.GetTypeString() << std::endl;
    else
        std::cout << "Perf: not timed, " << gemm.GetTypeString() << std::endl;

    if(do_verification)
    {
        auto ref     = ReferenceGemmInstance{};
        auto ref_inv = ref.MakeInvoker();
        ref_inv.Run(ref.MakeArgument(a_m_k, quant_b_k_n, scale_k_n, c_host, PassThrough{}, PassThrough{}, PassThrough{}));
        c_buf.FromDevice(c_device.mData.data());
        const bool ok = ck::utils::check_err(c_device, c_host);
        std::cout << (ok ? "verification: PASS" : "verification: FAIL") << std::endl;
        return ok ? 0 : 1;
    }
    return 0;
}
