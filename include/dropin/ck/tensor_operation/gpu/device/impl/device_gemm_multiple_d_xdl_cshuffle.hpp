// Overlay header: put include/dropin IN FRONT of a Composable Kernel checkout's include paths and the two XDL
// sources of example/14_gemm_quantization (gemm_xdl_quantization_int8.cpp, gemm_xdl_bias_relu_quantization_int8.cpp)
// build unmodified on MI355X -- their `DeviceGemmMultipleD_Xdl_CShuffle<int8 ...>` instantiations resolve to the
// gfx950 int8 matrix-core engine (CK's class: impl/device_gemm_multiple_d_xdl_cshuffle.hpp:103-160).
//
// The class template accepts CK's full parameter list so the instantiations compile as written; the tile and
// transfer parameters describe CK's kernel and are ignored (the engine picks its own tiles).  Only the int8
// requantisation forms are mapped: no D tensor with Activation_Mul_Clamp<Act>, or one int32 D tensor (the bias)
// with Add_Activation_Mul_Clamp<Act>; any other instantiation fails to compile at the selector below.
#pragma once

#include "ck/utility/sequence.hpp"
#include "ck/utility/loop_scheduler.hpp"
#include "ck/tensor_operation/gpu/grid/gridwise_gemm_pipeline_selector.hpp"
#include "ck/tensor_operation/gpu/device/gemm_specialization.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"

namespace ck {
namespace tensor_operation {
namespace device {

namespace wq_dropin {
template <typename ALayout, typename BLayout, typename ELayout, typename ADataType, typename BDataType,
          typename DsDataType, typename EDataType, typename CDEElementwiseOperation>
struct QuantizationOperator; // no definition: the instantiation is not an int8 requantisation GEMM

template <typename ALayout, typename BLayout, typename ELayout, typename Activation>
struct QuantizationOperator<ALayout, BLayout, ELayout, int8_t, int8_t, ck::Tuple<>, int8_t,
                            element_wise::Activation_Mul_Clamp<Activation>>
{
    using type = DeviceGemmQuantization_gfx950<ALayout, BLayout, ELayout, element_wise::Activation_Mul_Clamp<Activation>>;
};

template <typename ALayout, typename BLayout, typename ELayout, typename Activation>
struct QuantizationOperator<ALayout, BLayout, ELayout, int8_t, int8_t, ck::Tuple<int32_t>, int8_t,
                            element_wise::Add_Activation_Mul_Clamp<Activation>>
{
    using type = DeviceGemmBiasQuantization_gfx950<ALayout, BLayout, ELayout, Activation>;
};
} // namespace wq_dropin

template <typename ALayout, typename BLayout, typename DsLayout, typename ELayout, typename ADataType,
          typename BDataType, typename AccDataType, typename CShuffleDataType, typename DsDataType, typename EDataType,
          typename AElementwiseOperation, typename BElementwiseOperation, typename CDEElementwiseOperation,
          GemmSpecialization GemmSpec, index_t NumGemmKPrefetchStage, index_t BlockSize, index_t MPerBlock,
          index_t NPerBlock, index_t KPerBlock, index_t AK1, index_t BK1, index_t MPerXDL, index_t NPerXDL,
          index_t MXdlPerWave, index_t NXdlPerWave, typename ABlockTransferThreadClusterLengths_AK0_M_AK1,
          typename ABlockTransferThreadClusterArrangeOrder, typename ABlockTransferSrcAccessOrder,
          index_t ABlockTransferSrcVectorDim, index_t ABlockTransferSrcScalarPerVector,
          index_t ABlockTransferDstScalarPerVector_AK1, index_t ABlockLdsExtraM,
          typename BBlockTransferThreadClusterLengths_BK0_N_BK1, typename BBlockTransferThreadClusterArrangeOrder,
          typename BBlockTransferSrcAccessOrder, index_t BBlockTransferSrcVectorDim,
          index_t BBlockTransferSrcScalarPerVector, index_t BBlockTransferDstScalarPerVector_BK1,
          index_t BBlockLdsExtraN, index_t CShuffleMXdlPerWavePerShuffle, index_t CShuffleNXdlPerWavePerShuffle,
          typename CDEBlockTransferClusterLengths_MBlock_MPerBlock_NBlock_NPerBlock,
          index_t CDEBlockTransferScalarPerVector_NPerBlock, LoopScheduler LoopSched = make_default_loop_scheduler(),
          PipelineVersion PipelineVer = PipelineVersion::v1, typename ComputeDataType = EDataType>
struct DeviceGemmMultipleD_Xdl_CShuffle
    : public wq_dropin::QuantizationOperator<ALayout, BLayout, ELayout, ADataType, BDataType, DsDataType, EDataType,
                                             CDEElementwiseOperation>::type
{
};

} // namespace device
} // namespace tensor_operation
} // namespace ck
