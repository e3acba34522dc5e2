// Small-M ("decode") kernels of the weight-quantized GEMM engine: M <= 64 rows of activations against the
// whole weight matrix.  The work is one pass over the weights (0.5 or 1 byte each), so the bound is HBM
// bandwidth, not the matrix core: the kernels are built to keep many 16-byte weight loads in flight per CU
// and to touch every weight byte once.
//
//   grid  (N / (64 NB), 1, k_batch), 256 threads: wave w of a workgroup owns NB blocks of 16 columns starting at
//         64 NB bx + 16 NB w, and the K slice of its blockIdx.z.  (The library instantiates NB = 1: NB = 2 halves
//         what the workgroups together re-read of the activations and the LDS fragment reads per weight, and was
//         measured no faster at M = 32 and 64 -- those sizes are bound by dequantisation + MFMA issue.)
//   lane  (fr, fq) = (l & 15, l >> 4) streams column fr of its wave: one 16-byte load per "super-step" =
//         its CH = 32 (int4) or 16 (8-bit) consecutive k, [fq CH, fq CH + CH) of the super-step's 4 CH = 128 or
//         64 k; word s of the load feeds MFMA k-step s.  (The k order inside a super-step is free as long as the
//         activation fragment uses the same mapping: physical k of (s, fq, j) is fq CH + 8 s + j.)
//   A     the activations (they come from L2: M K 2 bytes, read by every workgroup) pass through LDS in chunks of
//         SPC super-steps, shared by the four waves: the chunk after the current one is fetched into registers
//         while the current one is multiplied, so the LDS footprint is one chunk (17 - 33 KiB) whatever K is and
//         several workgroups fit a CU -- that, times the P loads each lane keeps in flight, is the number of
//         weight bytes in flight that HBM latency asks for.
//   MFMA  v_mfma_f32_16x16x32 with the weight fragment first: lane holds C[m = 16 mb + fr][n = 4 fq + j]; a
//         dequantised weight fragment is reused for the up to four 16-row blocks of activations.
//   split-K across blockIdx.z gives the chip enough workgroups when N / 64 is small; the fp32 partial slabs are
//   summed by wq_reduce_finalize.
#pragma once

#include "wq_kernels.h"

namespace wq {

// Where lane (column `col`, k quarter fq) of a weight stream finds its 16 bytes of the super-step that starts at k = kbeg
// (a multiple of the super-step), and how far apart consecutive super-steps are.  Plain layouts: K runs along the
// column, 64 bytes per super-step.  Pre-shuffled layouts (p.b_preshuf): the lane's unit of super-step i is unit
// 4 i + fq of its column, i.e. k half fq % 2 of block 2 i + fq / 2 (packed int4: 2048 bytes per super-step) or k
// quarter fq of block i (int8: 1024 bytes per super-step) -- a wave-wide load then reads whole 128-byte lines.
template <int BKIND>
__device__ __forceinline__ const uint8_t* weight_stream(const GemmParams& p, int col, int kbeg, int fq, size_t& step_bytes)
{
    if(p.b_preshuf)
    {
        if constexpr(BKIND == B_I4P)
        {
            step_bytes = 2048;
            return (const uint8_t*)p.b + preshuf_unit_offset(col, (kbeg >> 5) + fq, p.ldb_bytes);
        }
        else
        {
            step_bytes = 1024;
            return (const uint8_t*)p.b + preshuf8_unit_offset(col, (kbeg >> 4) + fq, p.ldb_bytes);
        }
    }
    step_bytes = 64;
    return (const uint8_t*)p.b + (size_t)col * p.ldb_bytes + (size_t)kbeg / 8 * BInfo<BKIND>::kBytesPer8K + fq * 16;
}

// Group-wise scales in the small-M kernels (GROUPED instances; groups of a power of two of at least the
// 32 (packed int4) or 16 (int8) k a lane takes from one super-step, so that a lane's weights of a super-step share one
// scale): the scale of (column, super-step) travels with that super-step's weights -- fetched P super-steps ahead from
// the (L2-resident) scale rows and installed right before the dequantisation.
__device__ __forceinline__ uint16_t group_scale(const GemmParams& p, const uint16_t* col_scale, int k)
{
    return col_scale[(size_t)(k >> p.scale_group_shift) * p.ld_scale];
}

// NMB: 16-row blocks of activations computed (1, 2 or 4; rows beyond M are zero in LDS and never stored).
// NB:  16-column blocks per wave (1 or 2).
// P:   weight loads in flight per lane (4 or 1); the number of super-steps of a K slice must be a multiple of P,
//      which keeps the loop body free of data-dependent branches (so the compiler's wait counts leave P - 1 loads
//      in flight).  The planner picks slices that allow P = 4; P = 1 serves everything else.
template <int AK, int BKIND, int NMB, int P, int NB, bool GROUPED = false>
__global__ __launch_bounds__(256) void wq_gemm_skinny(const GemmParams p)
{
    constexpr int CH     = BKIND == B_I4P ? 32 : 16; // k per lane per super-step (16 bytes of weights)
    constexpr int KS     = 4 * CH;                   // k per super-step
    constexpr int STEPS  = CH / 8;                   // MFMA k-steps per super-step
    constexpr int SPC    = (NMB == 1 || P == 1) ? P : 2; // super-steps per activation chunk
    constexpr int KC     = SPC * KS;                 // k per activation chunk
    constexpr int ROWS   = NMB * 16;
    constexpr int PITCH  = KC * 2 + 16;              // bytes; the 16 spread the rows of a fragment read over the banks
    constexpr int CPR    = KC / 8;                   // 16-byte pieces per row of a chunk
    constexpr int PIECES = ROWS * CPR;
    constexpr int APT    = (PIECES + 255) / 256;     // pieces per thread
    __shared__ __attribute__((aligned(16))) uint8_t lds[ROWS * PITCH];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr   = lane & 15;
    const int fq   = lane >> 4;
    const int n0   = blockIdx.x * (64 * NB) + wave * (16 * NB); // column block nb of the wave starts at n0 + 16 nb

    const int kz_begin = p.k_begin + blockIdx.z * p.k_per_z;
    const int kz_end   = min(p.k_end, kz_begin + p.k_per_z);
    const int klen     = max(0, kz_end - kz_begin);
    const int nsteps   = klen / KS; // a multiple of P (host-checked)
    const int nchunks  = nsteps / SPC;

    f4_t acc[NB][NMB];
#pragma unroll
    for(int nb = 0; nb < NB; ++nb)
#pragma unroll
        for(int mb = 0; mb < NMB; ++mb)
            acc[nb][mb] = f4_t{0.f, 0.f, 0.f, 0.f};

    if(nsteps > 0)
    {
        // ---- weight stream: the first P loads go out before anything else ----------------------------------------
        const uint8_t* bptr[NB];
        size_t bstep = 64;
#pragma unroll
        for(int nb = 0; nb < NB; ++nb)
            bptr[nb] = weight_stream<BKIND>(p, n0 + 16 * nb + fr, kz_begin, fq, bstep);
        u4_t raw[P][NB];
#pragma unroll
        for(int i = 0; i < P; ++i)
#pragma unroll
            for(int nb = 0; nb < NB; ++nb)
                raw[i][nb] = *(const u4_t*)(bptr[nb] + (size_t)i * bstep);
        const uint16_t* sptr[NB]; // GROUPED: scale column of this lane's weight column
        uint16_t sraw[P][NB];
        if constexpr(GROUPED)
        {
#pragma unroll
            for(int nb = 0; nb < NB; ++nb)
                sptr[nb] = (const uint16_t*)p.scale + n0 + 16 * nb + fr;
#pragma unroll
            for(int i = 0; i < P; ++i)
#pragma unroll
                for(int nb = 0; nb < NB; ++nb)
                    sraw[i][nb] = group_scale(p, sptr[nb], kz_begin + i * KS + fq * CH);
        }

        // ---- activation chunks: piece q = tid + 256 j of a chunk is 16 bytes of row q / CPR ------------------------
        const uint8_t* a_src[APT];
        int a_dst[APT];
        bool a_ok[APT];
#pragma unroll
        for(int j = 0; j < APT; ++j)
        {
            const int q   = threadIdx.x + 256 * j;
            const int row = q / CPR, ch = q - row * CPR;
            a_ok[j]       = q < PIECES && row < p.M;
            a_src[j]      = (const uint8_t*)p.a + ((size_t)(a_ok[j] ? row : 0) * p.lda + kz_begin) * 2 + ch * 16;
            a_dst[j]      = q < PIECES ? row * PITCH + ch * 16 : -1;
        }
        u4_t apre[APT];
        auto fetch_chunk = [&](int c) __attribute__((always_inline)) {
#pragma unroll
            for(int j = 0; j < APT; ++j)
            {
                apre[j] = *(const u4_t*)(a_src[j] + (size_t)c * KC * 2);
                if(!a_ok[j])
                    apre[j] = u4_t{0u, 0u, 0u, 0u};
            }
        };
        fetch_chunk(0);

        ScaleReg<AK> sc[NB];
#pragma unroll
        for(int nb = 0; nb < NB; ++nb)
            sc[nb].set(GROUPED ? unit_scale_bits<AK>()
                               : (p.scale ? ((const uint16_t*)p.scale)[n0 + 16 * nb + fr] : unit_scale_bits<AK>()));

        const uint8_t* a_lane = lds + fr * PITCH + fq * CH * 2;
        for(int t = 0; t < nsteps; t += P)
        {
#pragma unroll
            for(int cc = 0; cc < P / SPC; ++cc)
            {
                const int c = t / SPC + cc;
                __syncthreads(); // every wave is done with the previous chunk
#pragma unroll
                for(int j = 0; j < APT; ++j)
                    if(a_dst[j] >= 0)
                        *(u4_t*)(lds + a_dst[j]) = apre[j];
                __syncthreads();
                // the next chunk travels while this one is multiplied (past the end: the last one again, unused)
                fetch_chunk(min(c + 1, nchunks - 1));
#pragma unroll
                for(int ii = 0; ii < SPC; ++ii)
                {
                    const int i   = cc * SPC + ii;
                    u4_t w[NB];
                    // prefetch super-step t + i + P; past the end re-load the last one (valid memory, value unused)
                    const int nxt = min(t + i + P, nsteps - 1);
#pragma unroll
                    for(int nb = 0; nb < NB; ++nb)
                    {
                        w[nb]      = raw[i][nb];
                        raw[i][nb] = *(const u4_t*)(bptr[nb] + (size_t)nxt * bstep);
                        if constexpr(GROUPED)
                        {
                            sc[nb].set(sraw[i][nb]);
                            sraw[i][nb] = group_scale(p, sptr[nb], kz_begin + nxt * KS + fq * CH);
                        }
                    }
                    u4_t af[STEPS][NMB];
#pragma unroll
                    for(int s = 0; s < STEPS; ++s)
#pragma unroll
                        for(int mb = 0; mb < NMB; ++mb)
                            af[s][mb] = *(const u4_t*)(a_lane + (ii * KS + 8 * s) * 2 + mb * 16 * PITCH);
#pragma unroll
                    for(int s = 0; s < STEPS; ++s)
#pragma unroll
                        for(int nb = 0; nb < NB; ++nb)
                        {
                            u4_t wf;
                            if constexpr(BKIND == B_I4P)
                                wf = dequant8(sc[nb], w[nb][s], std::integral_constant<int, B_I4P>{});
                            else
                            {
                                const u2_t w8 = {w[nb][2 * s], w[nb][2 * s + 1]};
                                if constexpr(AK == A_F16)
                                    wf = dequant8_bytes_f16<BKIND>(sc[nb], w8);
                                else
                                    wf = dequant8_bytes_bf16<BKIND>(sc[nb], w8);
                            }
#pragma unroll
                            for(int mb = 0; mb < NMB; ++mb)
                            {
                                if constexpr(AK == A_F16)
                                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                        __builtin_bit_cast(h8_t, wf), __builtin_bit_cast(h8_t, af[s][mb]), acc[nb][mb], 0, 0, 0);
                                else
                                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf8_t, wf), __builtin_bit_cast(bf8_t, af[s][mb]), acc[nb][mb], 0, 0, 0);
                            }
                        }
                }
            }
        }
    }

    // ---- store: lane holds C[16 mb + fr][n0 + 16 nb + 4 fq .. +3] -------------------------------------------------
#pragma unroll
    for(int nb = 0; nb < NB; ++nb)
#pragma unroll
        for(int mb = 0; mb < NMB; ++mb)
        {
            const int m = mb * 16 + fr;
            if(m < p.M)
            {
                const int n = n0 + 16 * nb + fq * 4;
                if(p.partial)
                    *(f4_t*)((float*)p.c + ((size_t)blockIdx.z * p.M + m) * p.ldc + n) = acc[nb][mb];
                else
                {
                    uint16_t o[4];
#pragma unroll
                    for(int j = 0; j < 4; ++j)
                        o[j] = cvt_out16(apply_epilogue(p.epi, acc[nb][mb][j], m, n + j), p.epi.c_dtype);
                    u2_t pk;
                    pk[0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
                    pk[1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
                    *(u2_t*)((uint16_t*)p.c + (size_t)m * p.ldc + n) = pk;
                }
            }
        }
}

// Variant for M <= 16 that needs neither split-K workspace nor a reduce kernel: the four waves of a workgroup
// own the SAME 16 columns and a quarter of K each; their accumulators are summed through LDS and wave 0 applies
// the epilogue.  grid (N / 16), 256 threads.  Only the M real rows of the activations are staged (lanes of the
// zero rows read nothing), so M = 1 costs 2 K bytes of LDS; the image must fit (the planner uses this variant up to
// 40 KiB).  K is split between the waves in units of P super-steps.
template <int AK, int BKIND, int P, bool GROUPED = false>
__global__ __launch_bounds__(256) void wq_gemm_skinny_ws(const GemmParams p)
{
    constexpr int CH    = BKIND == B_I4P ? 32 : 16;
    constexpr int KS    = 4 * CH;
    constexpr int STEPS = CH / 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr   = lane & 15;
    const int fq   = lane >> 4;
    const int n0   = blockIdx.x * 16;

    const int klen   = p.k_end - p.k_begin;
    // K is dealt to the four waves in units of P super-steps (the host guarantees klen % (P KS) == 0 and at least four
    // units); the first units % 4 waves take one unit more
    const int units  = klen / (KS * P);
    const int ubase  = units >> 2, urem = units & 3;
    const int nsteps = (ubase + (wave < urem ? 1 : 0)) * P;
    const int koff   = (wave * ubase + min(wave, urem)) * (P * KS); // first k of this wave, from k_begin
    const int kbeg   = p.k_begin + koff;

    size_t bstep        = 64;
    const uint8_t* bptr = weight_stream<BKIND>(p, n0 + fr, kbeg, fq, bstep);
    u4_t raw[P];
#pragma unroll
    for(int i = 0; i < P; ++i)
        raw[i] = *(const u4_t*)(bptr + (size_t)i * bstep);
    const uint16_t* sptr = (const uint16_t*)p.scale + n0 + fr; // GROUPED: scale column of this lane's weight column
    uint16_t sraw[P];
    if constexpr(GROUPED)
    {
#pragma unroll
        for(int i = 0; i < P; ++i)
            sraw[i] = group_scale(p, sptr, kbeg + i * KS + fq * CH);
    }

    // activations: rows 0 .. M-1 of [k_begin, k_end), row pitch klen * 2 + 16 bytes
    const int pitch = klen * 2 + 16;
    {
        const int chunks_per_row = klen / 8;
        const int total          = p.M * chunks_per_row;
        if(p.a_f8) // fp8 bytes: eight of them become the same 16 bytes of fp16 (every fp8 value is an fp16 value)
            for(int i = threadIdx.x; i < total; i += 256)
            {
                const int row  = i / chunks_per_row, ch = i - row * chunks_per_row;
                const u2_t raw = *(const u2_t*)((const uint8_t*)p.a + (size_t)row * p.lda + p.k_begin + ch * 8);
                u4_t o;
#pragma unroll
                for(int j = 0; j < 4; ++j)
                {
                    const uint32_t w = raw[j >> 1] >> (16 * (j & 1));
                    const uint8_t b0 = (uint8_t)(w & 0xFF), b1 = (uint8_t)((w >> 8) & 0xFF);
                    const _Float16 lo = (_Float16)(p.a_f8 == 2 ? f8_e4m3fn_to_f32(b0) : f8_fnuz_to_f32(b0));
                    const _Float16 hi = (_Float16)(p.a_f8 == 2 ? f8_e4m3fn_to_f32(b1) : f8_fnuz_to_f32(b1));
                    o[j] = (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
                }
                *(u4_t*)(lds + row * pitch + ch * 16) = o;
            }
        else
            for(int i = threadIdx.x; i < total; i += 256)
            {
                const int row = i / chunks_per_row, ch = i - row * chunks_per_row;
                *(u4_t*)(lds + row * pitch + ch * 16) =
                    *(const u4_t*)((const uint8_t*)p.a + ((size_t)row * p.lda + p.k_begin) * 2 + ch * 16);
            }
    }
    ScaleReg<AK> sc;
    sc.set(GROUPED ? unit_scale_bits<AK>() : (p.scale ? ((const uint16_t*)p.scale)[n0 + fr] : unit_scale_bits<AK>()));
    __syncthreads();

    f4_t acc            = {0.f, 0.f, 0.f, 0.f};
    const bool live     = fr < p.M; // this lane's activation row exists
    const uint8_t* a_ln = lds + (live ? fr : 0) * pitch + (koff + fq * CH) * 2;
    for(int t = 0; t < nsteps; t += P)
    {
#pragma unroll
        for(int i = 0; i < P; ++i)
        {
            const u4_t w  = raw[i];
            const int nxt = min(t + i + P, nsteps - 1);
            raw[i]        = *(const u4_t*)(bptr + (size_t)nxt * bstep);
            if constexpr(GROUPED)
            {
                sc.set(sraw[i]);
                sraw[i] = group_scale(p, sptr, kbeg + nxt * KS + fq * CH);
            }
            u4_t af[STEPS];
#pragma unroll
            for(int s = 0; s < STEPS; ++s)
            {
                af[s] = *(const u4_t*)(a_ln + ((t + i) * KS + 8 * s) * 2);
                if(!live)
                    af[s] = u4_t{0u, 0u, 0u, 0u};
            }
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
                if constexpr(AK == A_F16)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wf), __builtin_bit_cast(h8_t, af[s]),
                                                                 acc, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, wf),
                                                                  __builtin_bit_cast(bf8_t, af[s]), acc, 0, 0, 0);
            }
        }
    }

    // sum the four K quarters in ascending order through LDS (the activation image is no longer needed)
    __syncthreads();
    f4_t* red = (f4_t*)lds;
    red[wave * 64 + lane] = acc;
    __syncthreads();
    if(wave == 0 && live)
    {
        f4_t v = red[lane];
#pragma unroll
        for(int w = 1; w < 4; ++w)
        {
            const f4_t u = red[w * 64 + lane];
            v[0] += u[0], v[1] += u[1], v[2] += u[2], v[3] += u[3];
        }
        const int m = fr, n = n0 + fq * 4;
        uint16_t o[4];
#pragma unroll
        for(int j = 0; j < 4; ++j)
            o[j] = cvt_out16(apply_epilogue(p.epi, v[j], m, n + j), p.epi.c_dtype);
        u2_t pk;
        pk[0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        pk[1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        *(u2_t*)((uint16_t*)p.c + (size_t)m * p.ldc + n) = pk;
    }
}

// The same division of work for activation images too large to stage whole (M up to 16 at any K): each wave passes
// the activations of its own K quarter through a
// private LDS region chunk by chunk (next chunk prefetched into registers; only the M real rows are moved, and
// lanes of the zero rows read nothing), so there is no barrier in the loop and the footprint is 9 - 33 KiB whatever
// K and M are.  Super-steps per wave must be a multiple of P.
template <int AK, int BKIND, int P, bool GROUPED = false>
__global__ __launch_bounds__(256) void wq_gemm_skinny_wsc(const GemmParams p)
{
    constexpr int CH     = BKIND == B_I4P ? 32 : 16;
    constexpr int KS     = 4 * CH;
    constexpr int STEPS  = CH / 8;
    constexpr int SPC    = P >= 2 ? 2 : 1;       // super-steps per activation chunk
    constexpr int KC     = SPC * KS;
    constexpr int PITCH  = KC * 2 + 16;
    constexpr int CPR    = KC / 8;               // 16-byte pieces per row of a chunk
    constexpr int APT    = 16 * CPR / 64;        // pieces per lane (16 rows)
    constexpr int REGION = 16 * PITCH;
    static_assert(16 * CPR % 64 == 0 && 4 * REGION >= 4 * 64 * 16);
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * REGION];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr   = lane & 15;
    const int fq   = lane >> 4;
    const int n0   = blockIdx.x * 16;

    const int klen    = p.k_end - p.k_begin;
    // K is dealt to the four waves in units of P super-steps, as in wq_gemm_skinny_ws
    const int units   = klen / (KS * P);
    const int ubase   = units >> 2, urem = units & 3;
    const int nsteps  = (ubase + (wave < urem ? 1 : 0)) * P;
    const int nchunks = nsteps / SPC;
    const int kbeg    = p.k_begin + (wave * ubase + min(wave, urem)) * (P * KS);

    size_t bstep        = 64;
    const uint8_t* bptr = weight_stream<BKIND>(p, n0 + fr, kbeg, fq, bstep);
    u4_t raw[P];
#pragma unroll
    for(int i = 0; i < P; ++i)
        raw[i] = *(const u4_t*)(bptr + (size_t)i * bstep);
    const uint16_t* sptr = (const uint16_t*)p.scale + n0 + fr; // GROUPED: scale column of this lane's weight column
    uint16_t sraw[P];
    if constexpr(GROUPED)
    {
#pragma unroll
        for(int i = 0; i < P; ++i)
            sraw[i] = group_scale(p, sptr, kbeg + i * KS + fq * CH);
    }

    // activation chunks of this wave: piece q = lane + 64 j is 16 bytes of row q / CPR
    uint8_t* my = lds + wave * REGION;
    const uint8_t* a_src[APT];
    int a_dst[APT];
    bool a_ok[APT];
#pragma unroll
    for(int j = 0; j < APT; ++j)
    {
        const int q   = lane + 64 * j;
        const int row = q / CPR, ch = q - row * CPR;
        a_ok[j]       = row < p.M;
        a_src[j]      = (const uint8_t*)p.a + ((size_t)(a_ok[j] ? row : 0) * p.lda + kbeg) * 2 + ch * 16;
        a_dst[j]      = row * PITCH + ch * 16;
    }
    u4_t apre[APT];
    auto fetch_chunk = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for(int j = 0; j < APT; ++j)
            if(a_ok[j])
                apre[j] = *(const u4_t*)(a_src[j] + (size_t)c * KC * 2);
    };
    fetch_chunk(0);

    ScaleReg<AK> sc;
    sc.set(GROUPED ? unit_scale_bits<AK>() : (p.scale ? ((const uint16_t*)p.scale)[n0 + fr] : unit_scale_bits<AK>()));

    f4_t acc            = {0.f, 0.f, 0.f, 0.f};
    const bool live     = fr < p.M; // this lane's activation row exists
    const uint8_t* a_ln = my + (live ? fr : 0) * PITCH + fq * CH * 2;
    for(int t = 0; t < nsteps; t += P)
    {
#pragma unroll
        for(int cc = 0; cc < P / SPC; ++cc)
        {
            const int c = t / SPC + cc;
            // the fragment reads of the previous chunk have been consumed by its MFMAs; LDS serves one wave's
            // accesses in order, the wait publishes the writes to the wave's other lanes before they are read
#pragma unroll
            for(int j = 0; j < APT; ++j)
                if(a_ok[j])
                    *(u4_t*)(my + a_dst[j]) = apre[j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            fetch_chunk(min(c + 1, nchunks - 1));
#pragma unroll
            for(int ii = 0; ii < SPC; ++ii)
            {
                const int i   = cc * SPC + ii;
                const u4_t w  = raw[i];
                const int nxt = min(t + i + P, nsteps - 1);
                raw[i]        = *(const u4_t*)(bptr + (size_t)nxt * bstep);
                if constexpr(GROUPED)
                {
                    sc.set(sraw[i]);
                    sraw[i] = group_scale(p, sptr, kbeg + nxt * KS + fq * CH);
                }
                u4_t af[STEPS];
#pragma unroll
                for(int s = 0; s < STEPS; ++s)
                {
                    af[s] = *(const u4_t*)(a_ln + (ii * KS + 8 * s) * 2);
                    if(!live)
                        af[s] = u4_t{0u, 0u, 0u, 0u};
                }
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
                    if constexpr(AK == A_F16)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, wf), __builtin_bit_cast(h8_t, af[s]),
                                                                     acc, 0, 0, 0);
                    else
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, wf),
                                                                      __builtin_bit_cast(bf8_t, af[s]), acc, 0, 0, 0);
                }
            }
            // all fragment reads of this chunk are complete before the next chunk overwrites the region
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }

    // sum the four K quarters in ascending order through LDS (the activation images are no longer needed)
    __syncthreads();
    f4_t* red = (f4_t*)lds;
    red[wave * 64 + lane] = acc;
    __syncthreads();
    if(wave == 0 && live)
    {
        f4_t v = red[lane];
#pragma unroll
        for(int w = 1; w < 4; ++w)
        {
            const f4_t u = red[w * 64 + lane];
            v[0] += u[0], v[1] += u[1], v[2] += u[2], v[3] += u[3];
        }
        const int m = fr, n = n0 + fq * 4;
        uint16_t o[4];
#pragma unroll
        for(int j = 0; j < 4; ++j)
            o[j] = cvt_out16(apply_epilogue(p.epi, v[j], m, n + j), p.epi.c_dtype);
        u2_t pk;
        pk[0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        pk[1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        *(u2_t*)((uint16_t*)p.c + (size_t)m * p.ldc + n) = pk;
    }
}

} // namespace wq
