// Instance factory of the gfx950 fp8 x packed-int4 MultiplyMultiply GEMM for Composable Kernel's client API.
//
// CK's factory for the MultiplyMultiply epilogue
// (library/include/ck/library/tensor_operation_instance/gpu/gemm_multiply_multiply.hpp:150-215) is specialised on
// DsLayout = Tuple<Row, Col>, DsDataType = Tuple<F32, F32> and lists fp8 x fp8 -> bf16 only.  This header adds the
// more specialised factory for BDataType = ck::wq_int4x2_t (two offset-binary nibbles per byte, layout at
// wqgemm_pack_int4 in include/wqgemm.h) with fp8 activations of either encoding -- ck::f8_t (E4M3 FNUZ) or
// ck::wq_f8_e4m3fn_t (OCP E4M3) -- and fp16 / bf16 output, the same interface otherwise: D0 row-major per column,
// D1 column-major per row.  Link composable_kernel_amd/ck_instances/device_gemm_multiply_multiply_weight_quant_gfx950_instance.cpp.
// Needs a CK checkout on the include path (device_operation_instance_factory.hpp is CK's).
#pragma once

#include <memory>
#include <vector>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_multiple_d.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/library/tensor_operation_instance/device_operation_instance_factory.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"

namespace ck {
namespace tensor_operation {
namespace device {
namespace instance {

template <typename ADataType, typename EDataType>
using WqMultiplyMultiplyOp =
    DeviceGemmMultipleD<Row, Col, Tuple<Row, Col>, Row, ADataType, ck::wq_int4x2_t, Tuple<F32, F32>, EDataType, PassThrough,
                        PassThrough, ck::tensor_operation::element_wise::MultiplyMultiply>;

void add_device_gemm_multiply_multiply_weight_quant_gfx950_f8_i4_bf16_mk_nk_mn_instances(
    std::vector<std::unique_ptr<WqMultiplyMultiplyOp<F8, BF16>>>& instances);
void add_device_gemm_multiply_multiply_weight_quant_gfx950_f8_i4_f16_mk_nk_mn_instances(
    std::vector<std::unique_ptr<WqMultiplyMultiplyOp<F8, F16>>>& instances);
void add_device_gemm_multiply_multiply_weight_quant_gfx950_f8e4m3fn_i4_bf16_mk_nk_mn_instances(
    std::vector<std::unique_ptr<WqMultiplyMultiplyOp<ck::wq_f8_e4m3fn_t, BF16>>>& instances);
void add_device_gemm_multiply_multiply_weight_quant_gfx950_f8e4m3fn_i4_f16_mk_nk_mn_instances(
    std::vector<std::unique_ptr<WqMultiplyMultiplyOp<ck::wq_f8_e4m3fn_t, F16>>>& instances);

template <typename ADataType, typename EDataType>
struct DeviceOperationInstanceFactory<WqMultiplyMultiplyOp<ADataType, EDataType>>
{
    using DeviceOp = WqMultiplyMultiplyOp<ADataType, EDataType>;
    static auto GetInstances()
    {
        std::vector<std::unique_ptr<DeviceOp>> op_ptrs;
        if constexpr(is_same_v<ADataType, f8_t> && is_same_v<EDataType, bhalf_t>)
            add_device_gemm_multiply_multiply_weight_quant_gfx950_f8_i4_bf16_mk_nk_mn_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, f8_t> && is_same_v<EDataType, half_t>)
            add_device_gemm_multiply_multiply_weight_quant_gfx950_f8_i4_f16_mk_nk_mn_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, ck::wq_f8_e4m3fn_t> && is_same_v<EDataType, bhalf_t>)
            add_device_gemm_multiply_multiply_weight_quant_gfx950_f8e4m3fn_i4_bf16_mk_nk_mn_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, ck::wq_f8_e4m3fn_t> && is_same_v<EDataType, half_t>)
            add_device_gemm_multiply_multiply_weight_quant_gfx950_f8e4m3fn_i4_f16_mk_nk_mn_instances(op_ptrs);
        return op_ptrs;
    }
};

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
