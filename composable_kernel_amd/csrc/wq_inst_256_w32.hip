// Translation unit of the 32x32x16-MFMA variant of the 256-tile kernel (16-bit activations x packed int4).
#include <hip/hip_runtime.h>
#include <atomic>

#include "wqgemm.h"
#include "wq_gemm_256_w32.h"

namespace wq {

namespace {

template <int AK, bool STAMP>
int launch_v(const GemmParams& gp, dim3 grid, hipStream_t s)
{
    // more than 64 KiB of dynamic LDS must be opted into once per device
    static std::atomic<bool> opted_in[64];
    int dev                  = 0;
    if(hipGetDevice(&dev) != hipSuccess)
        return WQ_ERR_HIP;
    constexpr int lds_bytes = Mfma256W32::LDS_BYTES;
    if(dev >= 0 && dev < 64 && !opted_in[dev])
    {
        if(hipFuncSetAttribute((const void*)wq_gemm_mfma_256_w32<AK, STAMP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes) != hipSuccess)
            return WQ_ERR_HIP;
        opted_in[dev] = true;
    }
    hipLaunchKernelGGL((wq_gemm_mfma_256_w32<AK, STAMP>), grid, dim3(512), lds_bytes, s, gp);
    return WQ_OK;
}

} // namespace

int launch_256_w32(const GemmParams& gp, dim3 grid, hipStream_t s, int ak)
{
    if(gp.dbg)
        return ak == A_F16 ? launch_v<A_F16, true>(gp, grid, s) : launch_v<A_BF16, true>(gp, grid, s);
    return ak == A_F16 ? launch_v<A_F16, false>(gp, grid, s) : launch_v<A_BF16, false>(gp, grid, s);
}

} // namespace wq
