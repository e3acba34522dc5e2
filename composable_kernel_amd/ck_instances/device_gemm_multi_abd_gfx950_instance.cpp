// Instance registration of the gfx950 bf16 x int8 weight-quantized GEMM family for Composable Kernel's multi-ABD
// instance factory (library/include/ck/library/tensor_operation_instance/gpu/gemm_multi_abd.hpp:24-260): the free
// functions the factory specialisations call, with the names and signatures CK's own instance library defines them
// with (library/src/tensor_operation_instance/gpu/gemm_multi_abd/*.cpp), implemented on this engine.  Linking this
// file in place of CK's lets client_example/30_gemm_bf16Aint8B/*.cpp find the gfx950 operators unchanged.
//
// Not defined: ..._multiply_bf16_i8_bf16_mk_kn_mn_bias_v1_instances.  Its declaration (gemm_multi_abd.hpp:198-209)
// pairs ONE D layout with TWO D types, while the factory branch that calls it (gemm_multi_abd.hpp:585-625) passes a
// vector of operators with two D layouts, so that branch cannot be instantiated in the reference either.  The operator
// itself exists here -- DeviceGemmMultiABD_gfx950<Row, Row, Row, BF16, I8, BF16, false, MultiplyAdd> with the reference's
// bit-pattern MultiplyAdd<bhalf_t, float, bhalf_t, bhalf_t> (element_wise_operation.hpp:200-231) -- and is reached by
// direct instantiation.  Host code only.
#include <memory>
#include <vector>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_multiple_abd.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"

namespace ck {
namespace tensor_operation {
namespace device {
namespace instance {

using BF16 = ck::bhalf_t;
using I8   = int8_t;
using Row  = ck::tensor_layout::gemm::RowMajor;
using Col  = ck::tensor_layout::gemm::ColumnMajor;

using PassThrough         = ck::tensor_operation::element_wise::PassThrough;
using Multiply            = ck::tensor_operation::element_wise::Multiply;
using Add                 = ck::tensor_operation::element_wise::Add;
using FastGelu            = ck::tensor_operation::element_wise::FastGelu;
using AddFastGelu         = ck::tensor_operation::element_wise::AddFastGelu;
using MultiplyFastGelu    = ck::tensor_operation::element_wise::MultiplyFastGelu;
using MultiplyAddFastGelu = ck::tensor_operation::element_wise::MultiplyAddFastGelu;

// B = Multiply(B0 int8, B1 bf16 scale): Bs has two tensors of one layout
template <typename ALayout, typename BLayout, typename DsLayout, typename DsDataType, typename CDE>
using ScaledBPtrs = std::vector<std::unique_ptr<
    DeviceGemmMultipleABD<ck::Tuple<ALayout>, ck::Tuple<BLayout, BLayout>, DsLayout, Row, ck::Tuple<BF16>, ck::Tuple<I8, BF16>,
                          DsDataType, BF16, PassThrough, Multiply, CDE>>>;
// B = PassThrough(B0 int8), scale applied by the CDE operation
template <typename DsLayout, typename DsDataType, typename CDE>
using PlainBPtrs = std::vector<std::unique_ptr<DeviceGemmMultipleABD<ck::Tuple<Row>, ck::Tuple<Row>, DsLayout, Row, ck::Tuple<BF16>,
                                                                     ck::Tuple<I8>, DsDataType, BF16, PassThrough, PassThrough, CDE>>>;

#define WQ_SCALED_B_FAMILY(TAG, ALAYOUT, BLAYOUT)                                                                            \
    void add_device_gemm_xdl_multi_abd_bf16_i8_bf16_##TAG##_mn_bias_gelu_v1_instances(                                       \
        ScaledBPtrs<ALAYOUT, BLAYOUT, ck::Tuple<Row>, ck::Tuple<BF16>, AddFastGelu>& instances)                              \
    {                                                                                                                        \
        instances.push_back(std::make_unique<DeviceGemmMultiABD_gfx950<ALAYOUT, BLAYOUT, Row, BF16, I8, BF16, true, AddFastGelu>>()); \
    }                                                                                                                        \
    void add_device_gemm_xdl_multi_abd_bf16_i8_bf16_##TAG##_mn_bias_v1_instances(                                            \
        ScaledBPtrs<ALAYOUT, BLAYOUT, ck::Tuple<Row>, ck::Tuple<BF16>, Add>& instances)                                      \
    {                                                                                                                        \
        instances.push_back(std::make_unique<DeviceGemmMultiABD_gfx950<ALAYOUT, BLAYOUT, Row, BF16, I8, BF16, true, Add>>()); \
    }                                                                                                                        \
    void add_device_gemm_xdl_multi_abd_bf16_i8_bf16_##TAG##_mn_gelu_v1_instances(                                            \
        ScaledBPtrs<ALAYOUT, BLAYOUT, ck::Tuple<>, ck::Tuple<>, FastGelu>& instances)                                        \
    {                                                                                                                        \
        instances.push_back(std::make_unique<DeviceGemmMultiABD_gfx950<ALAYOUT, BLAYOUT, Row, BF16, I8, BF16, true, FastGelu>>()); \
    }                                                                                                                        \
    void add_device_gemm_xdl_multi_abd_bf16_i8_bf16_##TAG##_mn_v1_instances(                                                 \
        ScaledBPtrs<ALAYOUT, BLAYOUT, ck::Tuple<>, ck::Tuple<>, PassThrough>& instances)                                     \
    {                                                                                                                        \
        instances.push_back(std::make_unique<DeviceGemmMultiABD_gfx950<ALAYOUT, BLAYOUT, Row, BF16, I8, BF16, true, PassThrough>>()); \
    }

WQ_SCALED_B_FAMILY(mk_kn, Row, Row)
WQ_SCALED_B_FAMILY(mk_nk, Row, Col)
WQ_SCALED_B_FAMILY(km_kn, Col, Row)
#undef WQ_SCALED_B_FAMILY

void add_device_gemm_xdl_multi_abd_multiply_bf16_i8_bf16_mk_kn_mn_bias_gelu_v1_instances(
    PlainBPtrs<ck::Tuple<Row, Row>, ck::Tuple<BF16, BF16>, MultiplyAddFastGelu>& instances)
{
    instances.push_back(std::make_unique<DeviceGemmMultiABD_gfx950<Row, Row, Row, BF16, I8, BF16, false, MultiplyAddFastGelu>>());
}
void add_device_gemm_xdl_multi_abd_multiply_bf16_i8_bf16_mk_kn_mn_gelu_v1_instances(
    PlainBPtrs<ck::Tuple<Row>, ck::Tuple<BF16>, MultiplyFastGelu>& instances)
{
    instances.push_back(std::make_unique<DeviceGemmMultiABD_gfx950<Row, Row, Row, BF16, I8, BF16, false, MultiplyFastGelu>>());
}
void add_device_gemm_xdl_multi_abd_multiply_bf16_i8_bf16_mk_kn_mn_v1_instances(
    PlainBPtrs<ck::Tuple<Row>, ck::Tuple<BF16>, Multiply>& instances)
{
    instances.push_back(std::make_unique<DeviceGemmMultiABD_gfx950<Row, Row, Row, BF16, I8, BF16, false, Multiply>>());
}

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
