// Overlay header: put include/dropin IN FRONT of a Composable Kernel checkout's include paths and
// example/35_splitK_gemm/gemm_xdl_splitk_reduce_bf16A_i8B.cpp builds unmodified on MI355X -- its
// `DeviceGemm_Xdl_CShuffleV3R1<..., bhalf_t, int8_t, ...>` instantiation resolves to the gfx950 split-K operator
// (CK's class: impl/device_gemm_xdl_cshuffle_v3r1.hpp:28-700).
//
// The class template accepts CK's full parameter list so the instantiation compiles as written; the tile, transfer
// and pipeline parameters describe CK's kernel and are ignored (the engine picks its own tiles).  Only weight-
// quantized problems are mapped: 16-bit floating A with int8 / uint8 / packed-int4 B.  An instantiation with a
// floating-point B (the other sources of example/35) fails to compile at dtype_of<BDataType>.
#pragma once

#include "ck/utility/sequence.hpp"
#include "ck/utility/blkgemmpipe_scheduler.hpp"
#include "ck/tensor_operation/gpu/block/blockwise_gemm_pipeline_xdlops_selector.hpp"
#include "ck/tensor_operation/gpu/device/gemm_specialization.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"

namespace ck {
namespace tensor_operation {
namespace device {

template <typename ALayout, typename BLayout, typename DsLayout, typename CLayout, typename ADataType,
          typename BDataType, typename DsDataType, typename CDataType, typename GemmAccDataType,
          typename CShuffleDataType, typename AElementwiseOperation, typename BElementwiseOperation,
          typename CElementwiseOperation, GemmSpecialization GemmSpec, index_t BlockSize, index_t MPerBlock,
          index_t NPerBlock, index_t KPerBlock, index_t AK1, index_t BK1, index_t MPerXDL, index_t NPerXDL,
          index_t MXdlPerWave, index_t NXdlPerWave, typename ABlockTransferThreadClusterLengths_AK0_M_AK1,
          typename ABlockTransferThreadClusterArrangeOrder, typename ABlockTransferSrcAccessOrder,
          index_t ABlockTransferSrcVectorDim, index_t ABlockTransferSrcScalarPerVector,
          index_t ABlockTransferDstScalarPerVector_AK1, bool ABlockLdsExtraM,
          typename BBlockTransferThreadClusterLengths_BK0_N_BK1, typename BBlockTransferThreadClusterArrangeOrder,
          typename BBlockTransferSrcAccessOrder, index_t BBlockTransferSrcVectorDim,
          index_t BBlockTransferSrcScalarPerVector, index_t BBlockTransferDstScalarPerVector_BK1, bool BBlockLdsExtraN,
          index_t CShuffleMXdlPerWavePerShuffle, index_t CShuffleNXdlPerWavePerShuffle,
          typename CShuffleBlockTransferClusterLengths_MBlock_MPerBlock_NBlock_NPerBlock,
          index_t CShuffleBlockTransferScalarPerVector_NPerBlock,
          BlockGemmPipelineScheduler BlkGemmPipeSched = BlockGemmPipelineScheduler::Intrawave,
          BlockGemmPipelineVersion BlkGemmPipelineVer = BlockGemmPipelineVersion::v1,
          typename ReduceDataType = CDataType, typename ComputeTypeA = CDataType, typename ComputeTypeB = ComputeTypeA>
struct DeviceGemm_Xdl_CShuffleV3R1
    : public DeviceGemmV2R1_WeightQuant_gfx950<ALayout, BLayout, DsLayout, CLayout, ADataType, BDataType, DsDataType,
                                               CDataType, AElementwiseOperation, BElementwiseOperation,
                                               CElementwiseOperation>
{
};

} // namespace device
} // namespace tensor_operation
} // namespace ck
