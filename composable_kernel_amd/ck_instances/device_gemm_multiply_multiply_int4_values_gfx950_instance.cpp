// Instance registration of the gfx950 fp8 x int4 MultiplyMultiply GEMM for weights held one per byte in ck::int4_t:
// the functions declared in include/ck/library/tensor_operation_instance/gpu/gemm_multiply_multiply_int4_values_gfx950.hpp,
// standing where add_device_gemm_multiply_multiply_xdl_f8_f8_bf16_mk_nk_mn_*_instances of CK's instance library stand for
// fp8 x fp8 (library/include/ck/library/tensor_operation_instance/gpu/gemm_multiply_multiply.hpp:20-148).  FNUZ activations
// are registered without the no-special-encodings promise, so that instance is exact in every encoding (fp16 image of A in
// operator-owned scratch, fp16 matrix-core kernel); E4M3FN activations run on the fp8 matrix core.  Host code only.
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

using Row              = ck::tensor_layout::gemm::RowMajor;
using Col              = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough      = ck::tensor_operation::element_wise::PassThrough;
using MultiplyMultiply = ck::tensor_operation::element_wise::MultiplyMultiply;

template <typename A, typename E>
using ValuePtrs = std::vector<std::unique_ptr<DeviceGemmMultipleD<Row, Col, ck::Tuple<Row, Col>, Row, A, ck::int4_t,
                                                                  ck::Tuple<float, float>, E, PassThrough, PassThrough, MultiplyMultiply>>>;

template <typename A, typename E>
static void add(ValuePtrs<A, E>& instances)
{
    instances.push_back(
        std::make_unique<DeviceGemmMultiplyMultiply_Int4Values_gfx950<Row, Col, Row, A, ck::int4_t, E, MultiplyMultiply>>());
}

void add_device_gemm_multiply_multiply_int4_values_gfx950_f8_i4_bf16_mk_nk_mn_instances(ValuePtrs<ck::f8_t, ck::bhalf_t>& i) { add<ck::f8_t, ck::bhalf_t>(i); }
void add_device_gemm_multiply_multiply_int4_values_gfx950_f8_i4_f16_mk_nk_mn_instances(ValuePtrs<ck::f8_t, ck::half_t>& i) { add<ck::f8_t, ck::half_t>(i); }
void add_device_gemm_multiply_multiply_int4_values_gfx950_f8e4m3fn_i4_bf16_mk_nk_mn_instances(ValuePtrs<ck::wq_f8_e4m3fn_t, ck::bhalf_t>& i) { add<ck::wq_f8_e4m3fn_t, ck::bhalf_t>(i); }
void add_device_gemm_multiply_multiply_int4_values_gfx950_f8e4m3fn_i4_f16_mk_nk_mn_instances(ValuePtrs<ck::wq_f8_e4m3fn_t, ck::half_t>& i) { add<ck::wq_f8_e4m3fn_t, ck::half_t>(i); }

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
