// Instance registration of the gfx950 int8 x int8 requantising GEMM for Composable Kernel's instance factory.
//
// DeviceOperationInstanceFactory<DeviceGemmMultipleD<..., int8, int8, Tuple<>, int8, PassThrough, PassThrough,
// Activation_Mul_Clamp<PassThrough>>>::GetInstances()
// (library/include/ck/library/tensor_operation_instance/gpu/quantization/gemm_quantization.hpp:140-227) collects its
// instances by calling free functions that CK defines in its static libraries
// (library/src/tensor_operation_instance/gpu/quantization/gemm/device_gemm_quantization_xdl_c_shuffle_i8_i8_i8_*_instance.cpp).
// This file defines the same functions - same names, same signatures, hence the same mangled symbols - on top of
// this engine, so a client that links it instead of CK's instance library (client_example/09_quantization) finds
// the gfx950 operator through the unchanged factory.  Host code only.
#include <memory>
#include <vector>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_multiple_d.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"

namespace ck {
namespace tensor_operation {
namespace device {
namespace instance {

using Row         = ck::tensor_layout::gemm::RowMajor;
using Col         = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
using Empty_Tuple = ck::Tuple<>;

template <typename ALayout, typename BLayout, typename Activation>
using QuantGemmPtrs = std::vector<std::unique_ptr<
    DeviceGemmMultipleD<ALayout, BLayout, Empty_Tuple, Row, int8_t, int8_t, Empty_Tuple, int8_t, PassThrough, PassThrough,
                        ck::tensor_operation::element_wise::Activation_Mul_Clamp<Activation>>>>;

template <typename ALayout, typename BLayout, typename Activation>
static void add(QuantGemmPtrs<ALayout, BLayout, Activation>& instances)
{
    instances.push_back(std::make_unique<DeviceGemmQuantization_gfx950<
                            ALayout, BLayout, Row, ck::tensor_operation::element_wise::Activation_Mul_Clamp<Activation>>>());
}

// Layout(A, B, C) = [Col, Row, Row]
void add_device_gemm_quantization_xdl_c_shuffle_i8_i8_i8_km_kn_mn_instances(QuantGemmPtrs<Col, Row, PassThrough>& instances)
{
    add<Col, Row, PassThrough>(instances);
}
// Layout(A, B, C) = [Col, Col, Row]
void add_device_gemm_quantization_xdl_c_shuffle_i8_i8_i8_km_nk_mn_instances(QuantGemmPtrs<Col, Col, PassThrough>& instances)
{
    add<Col, Col, PassThrough>(instances);
}
// Layout(A, B, C) = [Row, Row, Row]
void add_device_gemm_quantization_xdl_c_shuffle_i8_i8_i8_mk_kn_mn_instances(QuantGemmPtrs<Row, Row, PassThrough>& instances)
{
    add<Row, Row, PassThrough>(instances);
}
// Layout(A, B, C) = [Row, Col, Row]
void add_device_gemm_quantization_xdl_c_shuffle_i8_i8_i8_mk_nk_mn_instances(QuantGemmPtrs<Row, Col, PassThrough>& instances)
{
    add<Row, Col, PassThrough>(instances);
}

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
