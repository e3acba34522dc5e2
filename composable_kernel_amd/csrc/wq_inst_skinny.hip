// Translation unit of the small-M kernel family.
#include <hip/hip_runtime.h>

#include "wqgemm.h"
#include "wq_gemm_skinny.h"

namespace wq {

namespace {

template <int AK, int BKIND, int NMB, int P>
int launch_t(const GemmParams& gp, dim3 grid, hipStream_t s)
{
    // group-wise scales: packed int4 or int8 weights
    if constexpr(BKIND != B_U8O)
        if(gp.scale_group_k > 0)
        {
            hipLaunchKernelGGL((wq_gemm_skinny<AK, BKIND, NMB, P, 1, true>), grid, dim3(256), 0, s, gp);
            return WQ_OK;
        }
    hipLaunchKernelGGL((wq_gemm_skinny<AK, BKIND, NMB, P, 1>), grid, dim3(256), 0, s, gp);
    return WQ_OK;
}

template <int AK, int BKIND>
int launch_ab(const GemmParams& gp, dim3 grid, hipStream_t s, int nmb, int p)
{
    if(nmb == 1)
        return p == 4 ? launch_t<AK, BKIND, 1, 4>(gp, grid, s) : launch_t<AK, BKIND, 1, 1>(gp, grid, s);
    if(nmb == 2)
        return p == 4 ? launch_t<AK, BKIND, 2, 4>(gp, grid, s) : launch_t<AK, BKIND, 2, 1>(gp, grid, s);
    return p == 4 ? launch_t<AK, BKIND, 4, 4>(gp, grid, s) : launch_t<AK, BKIND, 4, 1>(gp, grid, s);
}

template <int AK, int BKIND, int P>
int launch_ws_t(const GemmParams& gp, dim3 grid, int lds_bytes, hipStream_t s)
{
    if constexpr(BKIND != B_U8O)
        if(gp.scale_group_k > 0)
        {
            if(lds_bytes > 0)
                hipLaunchKernelGGL((wq_gemm_skinny_ws<AK, BKIND, P, true>), grid, dim3(256), lds_bytes, s, gp);
            else
                hipLaunchKernelGGL((wq_gemm_skinny_wsc<AK, BKIND, P, true>), grid, dim3(256), 0, s, gp);
            return WQ_OK;
        }
    if(lds_bytes > 0) // whole activation image in (dynamic) LDS
        hipLaunchKernelGGL((wq_gemm_skinny_ws<AK, BKIND, P>), grid, dim3(256), lds_bytes, s, gp);
    else // chunked, per-wave regions (static LDS)
        hipLaunchKernelGGL((wq_gemm_skinny_wsc<AK, BKIND, P>), grid, dim3(256), 0, s, gp);
    return WQ_OK;
}

template <int AK, int BKIND>
int launch_ws_ab(const GemmParams& gp, dim3 grid, int lds_bytes, hipStream_t s, int p)
{
    return p == 4   ? launch_ws_t<AK, BKIND, 4>(gp, grid, lds_bytes, s)
           : p == 2 ? launch_ws_t<AK, BKIND, 2>(gp, grid, lds_bytes, s)
                    : launch_ws_t<AK, BKIND, 1>(gp, grid, lds_bytes, s);
}

} // namespace

// waves-split-K variants (M <= 16): p = loads in flight, super-steps per wave % p == 0; lds_bytes > 0 selects the
// whole-image kernel (at most 40 KiB, below the 64 KiB that need no opt-in), 0 the chunked one
int launch_skinny_ws(int ak, int bk, const GemmParams& gp, dim3 grid, int lds_bytes, hipStream_t s, int p)
{
    if(ak == A_F16)
        return bk == B_I4P   ? launch_ws_ab<A_F16, B_I4P>(gp, grid, lds_bytes, s, p)
               : bk == B_U8O ? launch_ws_ab<A_F16, B_U8O>(gp, grid, lds_bytes, s, p)
                             : launch_ws_ab<A_F16, B_I8>(gp, grid, lds_bytes, s, p);
    return bk == B_I4P   ? launch_ws_ab<A_BF16, B_I4P>(gp, grid, lds_bytes, s, p)
           : bk == B_U8O ? launch_ws_ab<A_BF16, B_U8O>(gp, grid, lds_bytes, s, p)
                         : launch_ws_ab<A_BF16, B_I8>(gp, grid, lds_bytes, s, p);
}

// nmb: 1, 2 or 4 sixteen-row blocks of activations; p: 4 or 1 loads in flight (super-steps per slice % p == 0)
int launch_skinny(int ak, int bk, const GemmParams& gp, dim3 grid, hipStream_t s, int nmb, int p)
{
    if(ak == A_F16)
        return bk == B_I4P   ? launch_ab<A_F16, B_I4P>(gp, grid, s, nmb, p)
               : bk == B_U8O ? launch_ab<A_F16, B_U8O>(gp, grid, s, nmb, p)
                             : launch_ab<A_F16, B_I8>(gp, grid, s, nmb, p);
    return bk == B_I4P   ? launch_ab<A_BF16, B_I4P>(gp, grid, s, nmb, p)
           : bk == B_U8O ? launch_ab<A_BF16, B_U8O>(gp, grid, s, nmb, p)
                         : launch_ab<A_BF16, B_I8>(gp, grid, s, nmb, p);
}

} // namespace wq
