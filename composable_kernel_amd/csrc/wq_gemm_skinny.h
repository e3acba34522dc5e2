// Small-M ("decode") kernel of the weight-quantized GEMM engine: M <= 64 rows of activations against the
// whole weight matrix.  The work is one pass over the weights (0.5 or 1 byte each), so the bound is HBM
// bandwidth, not the matrix core: the kernel is built to keep many 16-byte weight loads in flight per wave
// and to touch every weight byte once.
//
//   grid  (N / 64, 1, k_batch), 256 threads: wave w of a workgroup owns the 16 columns 64 bx + 16 w .. +15 and
//         the K slice of its blockIdx.z; the activations of that slice (all <= 64 rows) are staged once per
//         workgroup in LDS and shared by the four waves.
//   lane  (fr, fq) = (l & 15, l >> 4) streams column fr of its wave: one 16-byte load per "super-step" =
//         its CH = 32 (int4) or 16 (8-bit) consecutive k, [fq CH, fq CH + CH) of the super-step's 4 CH = 128 or
//         64 k; word s of the load feeds MFMA k-step s.  (The k order inside a super-step is free as long as the
//         activation fragment uses the same mapping: physical k of (s, fq, j) is fq CH + 8 s + j.)
//   MFMA  v_mfma_f32_16x16x32 with the weight fragment first: lane holds C[m = 16 mb + fr][n = 4 fq + j]; a
//         dequantised weight fragment is reused for the up to four 16-row blocks of activations.
//   split-K across blockIdx.z gives the chip enough workgroups (a 4096 x 4096 int4 matrix is 32 KiB per CU); the
//   fp32 partial slabs are summed by wq_reduce_finalize.
#pragma once

#include "wq_kernels.h"

namespace wq {

template <int AK, int BKIND>
__global__ __launch_bounds__(256) void wq_gemm_skinny(const GemmParams p)
{
    constexpr int CH    = BKIND == B_I4P ? 32 : 16; // k per lane per super-step (16 bytes of weights)
    constexpr int KS    = 4 * CH;                   // k per super-step
    constexpr int STEPS = CH / 8;                   // MFMA k-steps per super-step
    constexpr int P     = 4;                        // weight loads in flight per lane
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr   = lane & 15;
    const int fq   = lane >> 4;
    const int n0   = blockIdx.x * 64 + wave * 16;

    const int kz_begin = p.k_begin + blockIdx.z * p.k_per_z;
    const int kz_end   = min(p.k_end, kz_begin + p.k_per_z);
    const int klen     = max(0, kz_end - kz_begin);
    const int nsteps   = klen / KS;
    const int nmb      = (p.M + 15) >> 4;

    // ---- weight stream: first loads go out before anything else ---------------------------------------------
    const uint8_t* bptr = (const uint8_t*)p.b + (size_t)(n0 + fr) * p.ldb_bytes +
                          (size_t)kz_begin / 8 * BInfo<BKIND>::kBytesPer8K + fq * 16;
    u4_t raw[P];
#pragma unroll
    for(int i = 0; i < P; ++i)
        if(i < nsteps)
            raw[i] = *(const u4_t*)(bptr + (size_t)i * 64);

    // ---- activations of this K slice -> LDS, rows >= M zero-filled; row pitch klen * 2 + 16 bytes ------------
    const int pitch = klen * 2 + 16;
    {
        const int chunks_per_row = klen / 8; // 16-byte chunks
        const int total          = nmb * 16 * chunks_per_row;
        for(int i = threadIdx.x; i < total; i += 256)
        {
            const int row = i / chunks_per_row, ch = i - row * chunks_per_row;
            u4_t v        = {0u, 0u, 0u, 0u};
            if(row < p.M)
                v = *(const u4_t*)((const uint8_t*)p.a + ((size_t)row * p.lda + kz_begin) * 2 + ch * 16);
            *(u4_t*)(lds + row * pitch + ch * 16) = v;
        }
    }
    ScaleReg<AK> sc;
    sc.set(p.scale ? ((const uint16_t*)p.scale)[n0 + fr] : unit_scale_bits<AK>());
    __syncthreads();

    f4_t acc[4];
#pragma unroll
    for(int mb = 0; mb < 4; ++mb)
        acc[mb] = f4_t{0.f, 0.f, 0.f, 0.f};

    const uint8_t* a_lane = lds + fr * pitch + fq * CH * 2;
    for(int t = 0; t < nsteps; t += P)
    {
#pragma unroll
        for(int i = 0; i < P; ++i)
        {
            if(t + i < nsteps) // wave-uniform
            {
                const u4_t w = raw[i];
                if(t + i + P < nsteps)
                    raw[i] = *(const u4_t*)(bptr + (size_t)(t + i + P) * 64);
#pragma unroll
                for(int s = 0; s < STEPS; ++s)
                {
                    u4_t wf;
                    if constexpr(BKIND == B_I4P)
                        wf = dequant8(sc, w[s], std::integral_constant<int, B_I4P>{});
                    else
                    {
                        const u2_t w8 = {w[2 * s], w[2 * s + 1]};
                        if constexpr(AK == A_F16)
                            wf = dequant8_bytes_f16<BKIND>(sc, w8);
                        else
                            wf = dequant8_bytes_bf16<BKIND>(sc, w8);
                    }
                    const uint8_t* a_step = a_lane + ((t + i) * KS + 8 * s) * 2;
#pragma unroll
                    for(int mb = 0; mb < 4; ++mb)
                    {
                        if(mb < nmb) // uniform
                        {
                            const u4_t af = *(const u4_t*)(a_step + mb * 16 * pitch);
                            if constexpr(AK == A_F16)
                                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wf),
                                                                                 __builtin_bit_cast(h8_t, af), acc[mb], 0, 0, 0);
                            else
                                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, wf),
                                                                                  __builtin_bit_cast(bf8_t, af), acc[mb], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- store: lane holds C[16 mb + fr][n0 + 4 fq .. +3] --------------------------------------------------------
#pragma unroll
    for(int mb = 0; mb < 4; ++mb)
    {
        const int m = mb * 16 + fr;
        if(mb < nmb && m < p.M)
        {
            const int n = n0 + fq * 4;
            if(p.partial)
                *(f4_t*)((float*)p.c + ((size_t)blockIdx.z * p.M + m) * p.ldc + n) = acc[mb];
            else
            {
                uint16_t o[4];
#pragma unroll
                for(int j = 0; j < 4; ++j)
                    o[j] = cvt_out16(apply_epilogue(p.epi, acc[mb][j], m, n + j), p.epi.c_dtype);
                u2_t pk;
                pk[0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
                pk[1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
                *(u2_t*)((uint16_t*)p.c + (size_t)m * p.ldc + n) = pk;
            }
        }
    }
}

} // namespace wq
