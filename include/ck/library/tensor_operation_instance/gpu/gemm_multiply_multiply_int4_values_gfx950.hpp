// Instance factory of the gfx950 fp8 x int4 MultiplyMultiply GEMM for weights held one per byte in CK's own int4 type.
//
// CK's factory for the MultiplyMultiply epilogue
// (library/include/ck/library/tensor_operation_instance/gpu/gemm_multiply_multiply.hpp:150-215) lists fp8 x fp8 and
// int8 x int8.  This header adds the more specialised factory for BDataType = ck::int4_t (the _BitInt(4) of
// include/ck/utility/data_type.hpp:10-14, one byte per element; needs CK_EXPERIMENTAL_BIT_INT_EXTENSION_INT4, CMake's
// USE_BITINT_EXTENSION_INT4) with fp8 activations of either encoding -- ck::f8_t (E4M3 FNUZ) or ck::wq_f8_e4m3fn_t
// (OCP E4M3) -- and fp16 / bf16 output: D0 row-major, D1 column-major, as CK's.  With it CK's unmodified profiling loop
// (profiler/include/profiler/profile_gemm_multiply_multiply_impl.hpp) generates, uploads and verifies int4 weights its own
// way and finds the gfx950 operator, which packs them two per byte before it runs.  Link
// composable_kernel_amd/ck_instances/device_gemm_multiply_multiply_int4_values_gfx950_instance.cpp.
#pragma once

#ifndef CK_EXPERIMENTAL_BIT_INT_EXTENSION_INT4
#error "ck::int4_t needs -DCK_EXPERIMENTAL_BIT_INT_EXTENSION_INT4 (CMake option USE_BITINT_EXTENSION_INT4)"
#endif

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
namespace element_wise {
// PassThrough for the two element types CK's list of explicit specialisations does not have
// (include/ck/tensor_operation/gpu/element/unary_element_wise_operation.hpp:28-225): the host ReferenceGemm widens every
// operand element through it (library/include/ck/library/reference_tensor_operation/cpu/reference_gemm.hpp:71-90).
template <>
__host__ __device__ inline void PassThrough::operator()<float, ck::int4_t>(float& y, const ck::int4_t& x) const
{
    y = static_cast<float>(static_cast<int>(x));
}
template <>
__host__ __device__ inline void PassThrough::operator()<float, ck::wq_f8_e4m3fn_t>(float& y, const ck::wq_f8_e4m3fn_t& x) const
{
    y = static_cast<float>(x);
}
} // namespace element_wise
namespace device {
namespace instance {

template <typename ADataType, typename EDataType>
using WqMultiplyMultiplyInt4ValuesOp =
    DeviceGemmMultipleD<Row, Col, Tuple<Row, Col>, Row, ADataType, ck::int4_t, Tuple<F32, F32>, EDataType, PassThrough,
                        PassThrough, ck::tensor_operation::element_wise::MultiplyMultiply>;

void add_device_gemm_multiply_multiply_int4_values_gfx950_f8_i4_bf16_mk_nk_mn_instances(
    std::vector<std::unique_ptr<WqMultiplyMultiplyInt4ValuesOp<F8, BF16>>>& instances);
void add_device_gemm_multiply_multiply_int4_values_gfx950_f8_i4_f16_mk_nk_mn_instances(
    std::vector<std::unique_ptr<WqMultiplyMultiplyInt4ValuesOp<F8, F16>>>& instances);
void add_device_gemm_multiply_multiply_int4_values_gfx950_f8e4m3fn_i4_bf16_mk_nk_mn_instances(
    std::vector<std::unique_ptr<WqMultiplyMultiplyInt4ValuesOp<ck::wq_f8_e4m3fn_t, BF16>>>& instances);
void add_device_gemm_multiply_multiply_int4_values_gfx950_f8e4m3fn_i4_f16_mk_nk_mn_instances(
    std::vector<std::unique_ptr<WqMultiplyMultiplyInt4ValuesOp<ck::wq_f8_e4m3fn_t, F16>>>& instances);

template <typename ADataType, typename EDataType>
struct DeviceOperationInstanceFactory<WqMultiplyMultiplyInt4ValuesOp<ADataType, EDataType>>
{
    using DeviceOp = WqMultiplyMultiplyInt4ValuesOp<ADataType, EDataType>;
    static auto GetInstances()
    {
        std::vector<std::unique_ptr<DeviceOp>> op_ptrs;
        if constexpr(is_same_v<ADataType, f8_t> && is_same_v<EDataType, bhalf_t>)
            add_device_gemm_multiply_multiply_int4_values_gfx950_f8_i4_bf16_mk_nk_mn_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, f8_t> && is_same_v<EDataType, half_t>)
            add_device_gemm_multiply_multiply_int4_values_gfx950_f8_i4_f16_mk_nk_mn_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, ck::wq_f8_e4m3fn_t> && is_same_v<EDataType, bhalf_t>)
            add_device_gemm_multiply_multiply_int4_values_gfx950_f8e4m3fn_i4_bf16_mk_nk_mn_instances(op_ptrs);
        else if constexpr(is_same_v<ADataType, ck::wq_f8_e4m3fn_t> && is_same_v<EDataType, half_t>)
            add_device_gemm_multiply_multiply_int4_values_gfx950_f8e4m3fn_i4_f16_mk_nk_mn_instances(op_ptrs);
        return op_ptrs;
    }
};

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
