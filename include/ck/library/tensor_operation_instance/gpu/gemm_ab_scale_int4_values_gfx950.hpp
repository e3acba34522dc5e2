// Instance factory of the gfx950 block-scaled fp8 x int4 GEMM for weights held one per byte in CK's own int4 type.
//
// CK's factory for DeviceGemmMultipleD_ABScale (library/include/ck/library/tensor_operation_instance/gpu/gemm_ab_scale.hpp)
// lists fp8 x fp8 -> bf16 with 128 x 128 x 128 scale blocks.  This header adds the more specialised factory for
// B0DataType = ck::int4_t (one byte per element; needs CK_EXPERIMENTAL_BIT_INT_EXTENSION_INT4) with fp8 activations of either
// encoding -- ck::f8_t (E4M3 FNUZ) or ck::wq_f8_e4m3fn_t (OCP E4M3) -- and bf16 / fp16 output, same layouts (mk, nk, mn) and
// block shape.  With it CK's unmodified profiling loop (profiler/include/profiler/profile_gemm_ab_scale_impl.hpp) generates,
// uploads and verifies int4 weights its own way and finds the gfx950 operators, which pack them two per byte before they run.
// Link composable_kernel_amd/ck_instances/device_gemm_ab_scale_int4_values_gfx950_instance.cpp.
#pragma once

#include <memory>
#include <vector>

// (the PassThrough specialisations for ck::int4_t / ck::wq_f8_e4m3fn_t operands of the host ReferenceGemm)
#include "ck/library/tensor_operation_instance/gpu/gemm_multiply_multiply_int4_values_gfx950.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_multiple_d_ab_scale.hpp"

namespace ck {
namespace tensor_operation {
namespace device {
namespace instance {

template <typename ADataType, typename EDataType>
using WqABScaleInt4ValuesOp = DeviceGemmMultipleD_ABScale<Row, Col, Tuple<>, Row, ADataType, F32, ck::int4_t, F32, Tuple<>, EDataType,
                                                          128, 128, 128, PassThrough, PassThrough, PassThrough>;

void add_device_gemm_ab_scale_int4_values_gfx950_f8_i4_bf16_mk_nk_mn_128_128_128_instances(
    std::vector<std::unique_ptr<WqABScaleInt4ValuesOp<F8, BF16>>>& instances);
void add_device_gemm_ab_scale_int4_values_gfx950_f8_i4_f16_mk_nk_mn_128_128_128_instances(
    std::vector<std::unique_ptr<WqABScaleInt4ValuesOp<F8, F16>>>& instances);
void add_device_gemm_ab_scale_int4_values_gfx950_f8e4m3fn_i4_bf16_mk_nk_mn_128_128_128_instances(
    std::vector<std::unique_ptr<WqABScaleInt4ValuesOp<ck::wq_f8_e4m3fn_t, BF16>>>& instances);
void add_device_gemm_ab_scale_int4_values_gfx950_f8e4m3fn_i4_f16_mk_nk_mn_128_128_128_instances(
    std::vector<std::unique_ptr<WqABScaleInt4ValuesOp<ck::wq_f8_e4m3fn_t, F16>>>& instances);

template <typename ADataType, typename EDataType>
struct DeviceOperationInstanceFactory<WqABScaleInt4ValuesOp<ADataType, EDataType>>
{
    using DeviceOp = WqABScaleInt4ValuesOp<ADataType, EDataType>;
    static auto GetInstances()
    {
        std::vector<std::unique_ptr<DeviceOp>> op_ptrs;
        if constexpr(is_same_v<ADataType, f8_t> && is_same_v<EDataType, bhalf_t>)
            add_device_gemm_ab_scale_int4_values_gfx950_f8_i4_bf16_mk_nk_mn_128_128_128_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, f8_t> && is_same_v<EDataType, half_t>)
            add_device_gemm_ab_scale_int4_values_gfx950_f8_i4_f16_mk_nk_mn_128_128_128_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, ck::wq_f8_e4m3fn_t> && is_same_v<EDataType, bhalf_t>)
            add_device_gemm_ab_scale_int4_values_gfx950_f8e4m3fn_i4_bf16_mk_nk_mn_128_128_128_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, ck::wq_f8_e4m3fn_t> && is_same_v<EDataType, half_t>)
            add_device_gemm_ab_scale_int4_values_gfx950_f8e4m3fn_i4_f16_mk_nk_mn_128_128_128_instances(op_ptrs);
        return op_ptrs;
    }
};

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
