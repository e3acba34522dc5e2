// Instance factory of the gfx950 split-K GEMM over weight-quantized B for Composable Kernel's client API.
//
// CK's own factory for DeviceGemmSplitK
// (library/include/ck/library/tensor_operation_instance/gpu/gemm_splitk.hpp:203-420) is one partial specialisation
// over all data types whose `if constexpr` ladder lists floating-point B only (f16, f32, f8 x f16, f16 x f8, bf16);
// with BDataType = int8_t it returns an empty list.  This header adds the more specialised factory for 16-bit
// floating A with int8 B -- the data types of example/35_splitK_gemm/gemm_xdl_splitk_reduce_bf16A_i8B.cpp -- so
// that CK's ckProfiler loop (profiler/include/profiler/profile_gemm_splitk_impl.hpp:101-207), instantiated with
// <bhalf_t, int8_t, float, bhalf_t, ...>, finds the gfx950 operator.  Include it next to (after or before) CK's
// gemm_splitk.hpp and link composable_kernel_amd/ck_instances/device_gemm_splitk_weight_quant_gfx950_instance.cpp.
// Needs a CK checkout on the include path (device_operation_instance_factory.hpp is CK's).
#pragma once

#include <memory>
#include <vector>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_splitk.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/library/tensor_operation_instance/device_operation_instance_factory.hpp"

namespace ck {
namespace tensor_operation {
namespace device {
namespace instance {

#define WQ_SPLITK_DECL(NAME, AL, BL, AT)                                                                         \
    void NAME(std::vector<std::unique_ptr<DeviceGemmSplitK<AL, BL, Row, AT, I8, AT, PassThrough, PassThrough,  \
                                                           PassThrough>>>& instances);
// names follow CK's scheme: <a>_<b>_<c> data types, then mk/km for A, kn/nk for B, mn for C
WQ_SPLITK_DECL(add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_mk_kn_mn_instances, Row, Row, BF16)
WQ_SPLITK_DECL(add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_mk_nk_mn_instances, Row, Col, BF16)
WQ_SPLITK_DECL(add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_mk_kn_mn_instances, Row, Row, F16)
WQ_SPLITK_DECL(add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_mk_nk_mn_instances, Row, Col, F16)
// column-major activations (the km_* layouts of library/include/ck/library/tensor_operation_instance/gpu/gemm_splitk.hpp):
// the general kernel runs them
WQ_SPLITK_DECL(add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_km_kn_mn_instances, Col, Row, BF16)
WQ_SPLITK_DECL(add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_km_nk_mn_instances, Col, Col, BF16)
WQ_SPLITK_DECL(add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_km_kn_mn_instances, Col, Row, F16)
WQ_SPLITK_DECL(add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_km_nk_mn_instances, Col, Col, F16)
#undef WQ_SPLITK_DECL

template <typename ADataType, typename ALayout, typename BLayout, typename CLayout, typename ComputeType>
struct DeviceOperationInstanceFactory<
    ck::tensor_operation::device::DeviceGemmSplitK<ALayout, BLayout, CLayout, ADataType, int8_t, ADataType,
                                                   ck::tensor_operation::element_wise::PassThrough,
                                                   ck::tensor_operation::element_wise::PassThrough,
                                                   ck::tensor_operation::element_wise::PassThrough, ComputeType>>
{
    using DeviceOp = DeviceGemmSplitK<ALayout, BLayout, CLayout, ADataType, int8_t, ADataType,
                                      ck::tensor_operation::element_wise::PassThrough,
                                      ck::tensor_operation::element_wise::PassThrough,
                                      ck::tensor_operation::element_wise::PassThrough, ComputeType>;

    static auto GetInstances()
    {
        std::vector<std::unique_ptr<DeviceOp>> op_ptrs;
        // (the default ComputeType is the C type: the only one the registration functions are declared for)
        if constexpr(is_same_v<ComputeType, ADataType> && is_same_v<ALayout, Row> && is_same_v<CLayout, Row>)
        {
            if constexpr(is_same_v<ADataType, bhalf_t> && is_same_v<BLayout, Row>)
                add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_mk_kn_mn_instances(op_ptrs);
            else if constexpr(is_same_v<ADataType, bhalf_t> && is_same_v<BLayout, Col>)
                add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_mk_nk_mn_instances(op_ptrs);
            else if constexpr(is_same_v<ADataType, half_t> && is_same_v<BLayout, Row>)
                add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_mk_kn_mn_instances(op_ptrs);
            else if constexpr(is_same_v<ADataType, half_t> && is_same_v<BLayout, Col>)
                add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_mk_nk_mn_instances(op_ptrs);
        }
        else if constexpr(is_same_v<ComputeType, ADataType> && is_same_v<ALayout, Col> && is_same_v<CLayout, Row>)
        {
            if constexpr(is_same_v<ADataType, bhalf_t> && is_same_v<BLayout, Row>)
                add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_km_kn_mn_instances(op_ptrs);
            else if constexpr(is_same_v<ADataType, bhalf_t> && is_same_v<BLayout, Col>)
                add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_km_nk_mn_instances(op_ptrs);
            else if constexpr(is_same_v<ADataType, half_t> && is_same_v<BLayout, Row>)
                add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_km_kn_mn_instances(op_ptrs);
            else if constexpr(is_same_v<ADataType, half_t> && is_same_v<BLayout, Col>)
                add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_km_nk_mn_instances(op_ptrs);
        }
        return op_ptrs;
    }
};

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
