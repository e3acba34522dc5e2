// Instance registration of the gfx950 split-K GEMM over int8 weights for Composable Kernel's instance factory:
// the functions declared in include/ck/library/tensor_operation_instance/gpu/gemm_splitk_weight_quant_gfx950.hpp,
// which that header's DeviceOperationInstanceFactory<DeviceGemmSplitK<..., int8_t, ...>> calls.  They stand where the
// add_device_gemm_xdl_splitk_* functions of CK's instance library stand for floating-point B
// (library/src/tensor_operation_instance/gpu/gemm_splitk/*.cpp).  Host code only.
#include <memory>
#include <vector>

#include "ck/ck.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/device/device_gemm_splitk.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"

namespace ck {
namespace tensor_operation {
namespace device {
namespace instance {

using Row         = ck::tensor_layout::gemm::RowMajor;
using Col         = ck::tensor_layout::gemm::ColumnMajor;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;

template <typename ALayout, typename BLayout, typename T>
using SplitKPtrs =
    std::vector<std::unique_ptr<DeviceGemmSplitK<ALayout, BLayout, Row, T, int8_t, T, PassThrough, PassThrough, PassThrough>>>;

template <typename ALayout, typename BLayout, typename T>
static void add(SplitKPtrs<ALayout, BLayout, T>& instances)
{
    instances.push_back(std::make_unique<DeviceGemmSplitK_WeightQuant_gfx950<ALayout, BLayout, Row, T, int8_t, T, PassThrough>>());
}

void add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_mk_kn_mn_instances(SplitKPtrs<Row, Row, ck::bhalf_t>& i) { add<Row, Row, ck::bhalf_t>(i); }
void add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_mk_nk_mn_instances(SplitKPtrs<Row, Col, ck::bhalf_t>& i) { add<Row, Col, ck::bhalf_t>(i); }
void add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_mk_kn_mn_instances(SplitKPtrs<Row, Row, ck::half_t>& i) { add<Row, Row, ck::half_t>(i); }
void add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_mk_nk_mn_instances(SplitKPtrs<Row, Col, ck::half_t>& i) { add<Row, Col, ck::half_t>(i); }
void add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_km_kn_mn_instances(SplitKPtrs<Col, Row, ck::bhalf_t>& i) { add<Col, Row, ck::bhalf_t>(i); }
void add_device_gemm_splitk_weight_quant_gfx950_bf16_i8_bf16_km_nk_mn_instances(SplitKPtrs<Col, Col, ck::bhalf_t>& i) { add<Col, Col, ck::bhalf_t>(i); }
void add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_km_kn_mn_instances(SplitKPtrs<Col, Row, ck::half_t>& i) { add<Col, Row, ck::half_t>(i); }
void add_device_gemm_splitk_weight_quant_gfx950_f16_i8_f16_km_nk_mn_instances(SplitKPtrs<Col, Col, ck::half_t>& i) { add<Col, Col, ck::half_t>(i); }

} // namespace instance
} // namespace device
} // namespace tensor_operation
} // namespace ck
