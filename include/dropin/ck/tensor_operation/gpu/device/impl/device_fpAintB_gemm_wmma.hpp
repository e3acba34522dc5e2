// Overlay header: put include/dropin IN FRONT of a Composable Kernel checkout's include paths and
// example/64_fpAintB_gemm builds unmodified on MI355X -- its `DeviceFpAintBGemm_Wmma_CShuffle<...>`
// instantiation resolves to the gfx950 MFMA engine instead of the gfx11/gfx12-only WMMA kernel
// (include/ck/tensor_operation/gpu/device/impl/device_fpAintB_gemm_wmma.hpp:30-110 in CK).
//
// The class template below accepts CK's full parameter list so the example's instantiation
// compiles as written; the tile/transfer tuning parameters describe the WMMA kernel and are
// ignored (the gfx950 engine picks its own tiles).
#pragma once

#include "ck/utility/loop_scheduler.hpp"
#include "ck/tensor_operation/gpu/grid/gridwise_gemm_pipeline_selector.hpp"
#include "ck/tensor_operation/gpu/device/gemm_specialization.hpp"
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"

namespace ck {
namespace tensor_operation {
namespace device {

template <typename ALayout, typename BLayout, typename CLayout, typename ADataType, typename BDataType,
          typename ScaleDataType, typename CDataType, typename AccDataType, typename CShuffleDataType,
          typename AElementwiseOperation, typename BElementwiseOperation, typename CElementwiseOperation,
          GemmSpecialization GemmSpec, ck::index_t NumPrefetch, ck::index_t BlockSize, ck::index_t MPerBlock,
          ck::index_t NPerBlock, ck::index_t KPerBlock, ck::index_t K1, ck::index_t MPerWmma,
          ck::index_t NPerWmma, ck::index_t MRepeat, ck::index_t NRepeat, typename ABlockTransferClusterLengths,
          typename ABlockTransferClusterArrangeOrder, typename ABlockTransferSrcAccessOrder,
          ck::index_t ABlockTransferSrcVectorDim, ck::index_t ABlockTransferSrcScalarPerVector,
          ck::index_t ABlockTransferDstScalarPerVector_K1, bool ABlockLdsAddExtraM,
          typename BBlockTransferClusterLengths, typename BBlockTransferClusterArrangeOrder,
          typename BBlockTransferSrcAccessOrder, ck::index_t BBlockTransferSrcVectorDim,
          ck::index_t BBlockTransferSrcScalarPerVector, ck::index_t BBlockTransferDstScalarPerVector_K1,
          bool BBlockLdsAddExtraN, index_t CShuffleMRepeatPerShuffle, index_t CShuffleNRepeatPerShuffle,
          typename CShuffleBlockTransferClusterLengths, index_t CShuffleBlockTransferScalarPerVector,
          ck::LoopScheduler LoopSched = make_default_loop_scheduler(),
          ck::PipelineVersion PipelineVer = ck::PipelineVersion::weight_only>
struct DeviceFpAintBGemm_Wmma_CShuffle
    : public DeviceFpAintBGemm_Mfma_gfx950<ALayout, BLayout, CLayout, ADataType, BDataType, ScaleDataType,
                                           CDataType, AElementwiseOperation, BElementwiseOperation,
                                           CElementwiseOperation>
{
};

} // namespace device
} // namespace tensor_operation
} // namespace ck
