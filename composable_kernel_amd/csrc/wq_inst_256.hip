// One translation unit per (activation type, weight format) pair of the 256-tile kernel family:
// compiled with -DWQ_INST_AK=<0|1> -DWQ_INST_BK=<0|1|2>.
#include <hip/hip_runtime.h>
#include <atomic>

#include "wqgemm.h"
#include "wq_gemm_256.h"

#define WQ_CAT_(a, b, c) launch_256_##a##_##b
#define WQ_CAT(a, b) WQ_CAT_(a, b, )

namespace wq {

namespace {

template <int AK, int BKIND, int WM, int SCHED, bool STAMP, bool GROUPED = false>
int launch_v(const GemmParams& gp, dim3 grid, hipStream_t s)
{
    // more than 64 KiB of dynamic LDS must be opted into once per device
    static std::atomic<bool> opted_in[64];
    int dev                  = 0;
    if(hipGetDevice(&dev) != hipSuccess)
        return WQ_ERR_HIP;
    constexpr int lds_bytes = Mfma256<BKIND>::LDS_BYTES + (GROUPED ? kGroupScaleLds : 0);
    if(dev >= 0 && dev < 64 && !opted_in[dev])
    {
        if(hipFuncSetAttribute((const void*)wq_gemm_mfma_256<AK, BKIND, WM, SCHED, STAMP, GROUPED>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return WQ_ERR_HIP;
        opted_in[dev] = true;
    }
    hipLaunchKernelGGL((wq_gemm_mfma_256<AK, BKIND, WM, SCHED, STAMP, GROUPED>), grid, dim3(512), lds_bytes, s, gp);
    return WQ_OK;
}

} // namespace

// variant = tuning bits of wqgemm_problem.flags (WQ_FLAG_LOCKSTEP, WQ_FLAG_WAVES_2X4, WQ_FLAG_DEQUANT_IN_L)
int WQ_CAT(WQ_INST_AK, WQ_INST_BK)(const GemmParams& gp, dim3 grid, hipStream_t s, int variant)
{
    constexpr int AK = WQ_INST_AK, BK = WQ_INST_BK;
    const bool w2x4 = variant & WQ_FLAG_WAVES_2X4;
#if WQ_INST_AK == 0 && WQ_INST_BK == 0
    if(gp.dbg) // diagnostic stamped builds (default wave grid only)
#ifdef WQ_INST_MINIMAL
        return launch_v<AK, BK, 1, 5, true>(gp, grid, s);
#else
        return (variant & WQ_FLAG_DEQUANT_IN_L) ? launch_v<AK, BK, 1, 2, true>(gp, grid, s)
                                                 : launch_v<AK, BK, 1, 5, true>(gp, grid, s);
#endif
#endif
#if !defined(WQ_INST_MINIMAL)
    if(gp.scale_group_k > 0) // group-wise scales: the default schedule and wave grid only
        return launch_v<AK, BK, 1, 5, false, true>(gp, grid, s);
#endif
#ifdef WQ_INST_MINIMAL // tuning builds: the default schedule only (a third of the compile time)
    return launch_v<AK, BK, 1, 5, false>(gp, grid, s);
#endif
    if(variant & WQ_FLAG_LOCKSTEP)
        return launch_v<AK, BK, 2, 0, false>(gp, grid, s);
    if(variant & WQ_FLAG_DEQUANT_IN_L)
        return w2x4 ? launch_v<AK, BK, 2, 2, false>(gp, grid, s) : launch_v<AK, BK, 1, 2, false>(gp, grid, s);
    return w2x4 ? launch_v<AK, BK, 2, 5, false>(gp, grid, s) : launch_v<AK, BK, 1, 5, false>(gp, grid, s);
}

} // namespace wq
