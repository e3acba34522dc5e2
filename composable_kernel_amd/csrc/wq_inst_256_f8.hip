// Translation unit of the fp8 x packed-int4 256-tile kernel.
#include <hip/hip_runtime.h>
#include <atomic>

#include "wqgemm.h"
#include "wq_gemm_256_f8.h"
#include "wq_gemm_256_f8s.h"

namespace wq {

namespace {

// more than 64 KiB of dynamic LDS must be opted into once per device
template <typename K>
int launch_big_lds(K kernel, int lds_bytes, std::atomic<bool> (&opted_in)[64], const GemmParams& gp, dim3 grid, hipStream_t s)
{
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess)
        return WQ_ERR_HIP;
    if(dev >= 0 && dev < 64 && !opted_in[dev])
    {
        if(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return WQ_ERR_HIP;
        opted_in[dev] = true;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(512), lds_bytes, s, gp);
    return WQ_OK;
}

} // namespace

int launch_256_f8(const GemmParams& gp, dim3 grid, hipStream_t s, int variant)
{
    static std::atomic<bool> opted_s[64], opted_pp[64], opted_stamp[64], opted_ab[64], opted_abr[64];
    if(gp.sa) // block-wise scales of both operands: activation scales per 128-row half of a tile, or per row
        return gp.bm % 128 == 0 ? launch_big_lds(wq_gemm_f8_mfma_256_abscale, Mfma256F8S::LDS_BYTES_ABSCALE, opted_ab, gp, grid, s)
                                : launch_big_lds(wq_gemm_f8_mfma_256_abscale_rows, Mfma256F8S::LDS_BYTES_ABSCALE, opted_abr, gp, grid, s);
    if(gp.dbg) // diagnostic: the stamped build of the ping-pong form
        return launch_big_lds(wq_gemm_f8_mfma_256_pp<true>, Mfma256F8::LDS_BYTES, opted_stamp, gp, grid, s);
    if(variant & WQ_FLAG_PINGPONG)
        return launch_big_lds(wq_gemm_f8_mfma_256_pp<false>, Mfma256F8::LDS_BYTES, opted_pp, gp, grid, s);
    return launch_big_lds(wq_gemm_f8_mfma_256, Mfma256F8S::LDS_BYTES, opted_s, gp, grid, s);
}

} // namespace wq
