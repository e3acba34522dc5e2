// Instance registration of the gfx950 block-scaled fp8 x int4 GEMM for weights held one per byte in ck::int4_t: the functions
// declared in include/ck/library/tensor_operation_instance/gpu/gemm_ab_scale_int4_values_gfx950.hpp, standing where
// add_device_gemm_ab_scale_xdl_f8_f8_bf16_mk_nk_mn_128_128_128_*_instances of CK's instance library stand for fp8 x fp8
// (library/include/ck/library/tensor_operation_instance/gpu/gemm_ab_scale.hpp).  FNUZ activations get two instances: one
// without the no-special-encodings promise (exact for every byte, on the general kernel) and one with it (the fp8 matrix
// core; CK's generators produce none of the three special bytes); E4M3FN activations run on the matrix core.  Host code only.
#include <memory>
#include <vector>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_multiple_d_ab_scale.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"

namespace ck {
namespace tensor_operation {
namespace device {
namespace instance {

using Row         = ck::tensor_layout::gemm::RowMajor;
using Col         = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;

template <typename A, typename E>
using ValuePtrs = std::vector<std::unique_ptr<DeviceGemmMultipleD_ABScale<Row, Col, ck::Tuple<>, Row, A, float, ck::int4_t, float,
                                                                          ck::Tuple<>, E, 128, 128, 128, PassThrough, PassThrough, PassThrough>>>;

template <typename A, typename E>
static void add(ValuePtrs<A, E>& instances)
{
    instances.push_back(std::make_unique<DeviceGemmMultiD_ABScale_Int4Values_gfx950<Row, Col, Row, A, ck::int4_t, E, 128, 128, 128, false>>());
    if constexpr(std::is_same_v<A, ck::f8_t>)
        instances.push_back(std::make_unique<DeviceGemmMultiD_ABScale_Int4Values_gfx950<Row, Col, Row, A, ck::int4_t, E, 128, 128, 128, true>>());
}

void add_device_gemm_ab_scale_int4_values_gfx950_f8_i4_bf16_mk_nk_mn_128_128_128_instances(ValuePtrs<ck::f8_t, ck::bhalf_t>& i) { add<ck::f8_t, ck::bhalf_t>(i); }
void add_device_gemm_ab_scale_int4_values_gfx950_f8_i4_f16_mk_nk_mn_128_128_128_instances(ValuePtrs<ck::f8_t, ck::half_t>& i) { add<ck::f8_t, ck::half_t>(i); }
void add_device_gemm_ab_scale_int4_values_gfx950_f8e4m3fn_i4_bf16_mk_nk_mn_128_128_128_instances(ValuePtrs<ck::wq_f8_e4m3fn_t, ck::bhalf_t>& i) { add<ck::wq_f8_e4m3fn_t, ck::bhalf_t>(i); }
void add_device_gemm_ab_scale_int4_values_gfx950_f8e4m3fn_i4_f16_mk_nk_mn_128_128_128_instances(ValuePtrs<ck::wq_f8_e4m3fn_t, ck::half_t>& i) { add<ck::wq_f8_e4m3fn_t, ck::half_t>(i); }

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
