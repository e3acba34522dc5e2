// Translation unit of the fp8 x packed-int4 256-tile kernel.
#include <hip/hip_runtime.h>

#include "wqgemm.h"
#include "wq_gemm_256_f8.h"

namespace wq {

namespace {

template <bool STAMP>
int launch_v(const GemmParams& gp, dim3 grid, hipStream_t s)
{
    // more than 64 KiB of dynamic LDS must be opted into once per device
    static bool opted_in[64] = {};
    int dev                  = 0;
    if(hipGetDevice(&dev) != hipSuccess)
        return WQ_ERR_HIP;
    constexpr int lds_bytes = Mfma256F8::LDS_BYTES;
    if(dev >= 0 && dev < 64 && !opted_in[dev])
    {
        if(hipFuncSetAttribute((const void*)wq_gemm_f8_mfma_256<STAMP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes) != hipSuccess)
            return WQ_ERR_HIP;
        opted_in[dev] = true;
    }
    hipLaunchKernelGGL((wq_gemm_f8_mfma_256<STAMP>), grid, dim3(512), lds_bytes, s, gp);
    return WQ_OK;
}

} // namespace

int launch_256_f8(const GemmParams& gp, dim3 grid, hipStream_t s, int variant)
{
    (void)variant;
    if(gp.dbg)
        return launch_v<true>(gp, grid, s);
    return launch_v<false>(gp, grid, s);
}

} // namespace wq
