// Link-level check of the instance registrations this engine adds to Composable Kernel's factory: every
// DeviceOperationInstanceFactory<...>::GetInstances() the two weight-quant factory headers specialise must resolve
// to the add_device_* functions of composable_kernel_amd/ck_instances/ and return one gfx950 operator.  Touches no
// GPU (operators are stateless objects).  Builds only over a CK checkout (examples/Makefile: dropin).
#include <cstdio>

#include "ck/library/tensor_operation_instance/gpu/gemm_splitk_weight_quant_gfx950.hpp"
#include "ck/library/tensor_operation_instance/gpu/gemm_multiply_multiply_weight_quant_gfx950.hpp"
#include "ck/library/tensor_operation_instance/gpu/gemm_splitk.hpp"
#include "ck/library/tensor_operation_instance/gpu/gemm_multiply_multiply.hpp"

using namespace ck::tensor_operation::device;
using namespace ck::tensor_operation::device::instance;
using MultiplyMultiply = ck::tensor_operation::element_wise::MultiplyMultiply;

template <typename DeviceOp>
static int report(const char* what)
{
    const auto ops = DeviceOperationInstanceFactory<DeviceOp>::GetInstances();
    for(const auto& op : ops)
        printf("%s: %s\n", what, op->GetTypeString().c_str());
    return static_cast<int>(ops.size());
}

int main()
{
    int n = 0;
    n += report<DeviceGemmSplitK<Row, Row, Row, BF16, I8, BF16, PassThrough, PassThrough, PassThrough>>("splitk bf16 i8 mk_kn");
    n += report<DeviceGemmSplitK<Row, Col, Row, BF16, I8, BF16, PassThrough, PassThrough, PassThrough>>("splitk bf16 i8 mk_nk");
    n += report<DeviceGemmSplitK<Row, Row, Row, F16, I8, F16, PassThrough, PassThrough, PassThrough>>("splitk f16 i8 mk_kn");
    n += report<DeviceGemmSplitK<Row, Col, Row, F16, I8, F16, PassThrough, PassThrough, PassThrough>>("splitk f16 i8 mk_nk");
    n += report<WqMultiplyMultiplyOp<F8, BF16>>("multiply_multiply f8 i4 bf16");
    n += report<WqMultiplyMultiplyOp<F8, F16>>("multiply_multiply f8 i4 f16");
    n += report<WqMultiplyMultiplyOp<ck::wq_f8_e4m3fn_t, BF16>>("multiply_multiply f8e4m3fn i4 bf16");
    n += report<WqMultiplyMultiplyOp<ck::wq_f8_e4m3fn_t, F16>>("multiply_multiply f8e4m3fn i4 f16");
    printf("found %d instances\n", n);
    return n == 8 ? 0 : 1;
}
