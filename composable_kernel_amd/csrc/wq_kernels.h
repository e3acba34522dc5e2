// gfx950 kernels of the weight-quantized GEMM engine.
//
//   wq_gemm_mfma_128      128x128x64 tile, 4 waves, LDS-DMA staging, in-register dequant,
//                         v_mfma_f32_16x16x32_{f16,bf16}
//   wq_gemm_generic       any shape / stride; one output element per thread (edge shapes, tests)
//   wq_reduce_finalize    split-K: sum fp32 partials, epilogue, convert
//   wq_pack_int4          int8-held int4 (K,N) column-major -> packed offset-binary nibbles
//
// Orientation of the MFMA: the WEIGHT fragment is the first operand (its 16 rows are 16 output
// channels n) and the ACTIVATION fragment the second (its 16 columns are 16 rows m of C), so the
// accumulator of lane l holds C[m = l & 15][n = 4 (l >> 4) + j], j = 0..3: four consecutive n of
// one row of a row-major C, i.e. one 8-byte (16-bit C) or 16-byte (fp32 partial) store.
#pragma once

#include "wq_device.h"

namespace wq {

struct GemmParams
{
    const void* a;     // activations, row-major (M, lda)
    const void* b;     // weights, K contiguous per column n
    const void* scale; // N scales of A's 16-bit type
    void* c;           // output, row-major (M, ldc); fp32 partial slabs when partial != 0
    int M, N, K;
    int lda;       // elements
    int ldb_bytes; // bytes between consecutive columns of b
    int ldc;       // elements
    int k_begin;   // first k of this launch
    int k_end;     // one past the last k of this launch
    int k_per_z;   // K extent handled by one blockIdx.z (split-K inside the GPU)
    int partial;   // 1: store raw fp32 accumulators to slab blockIdx.z of c (ld = ldc)
    int tiles_m, tiles_n;
    int group_m;      // tile rows walked together by neighbouring workgroups (L2 / fabric locality)
    int direct_store; // 256-tile kernel: store accumulator fragments directly (tuning A/B)
    int a_scale;      // fp8 kernel: E8M0 block scale of the activation operand, replicated in all 4 bytes
    void* dbg;        // diagnostic builds only: per-wave cycle sums
    EpilogueArgs epi;
    int scale_group_k; // > 0: group-wise scales, scale[(k / scale_group_k) * ld_scale + n]
    int ld_scale;
    int scale_group_shift; // log2(scale_group_k) when that is a power of two (small-M kernels), else -1
    int a_f8;          // small-M whole-image kernel only: a holds fp8 bytes (1: E4M3 FNUZ, 2: OCP E4M3FN; lda in bytes), widened
                       // to fp16 (exact) on their way into LDS; 0: a holds the 16-bit activations
    // fp8 kernel, block-wise scales of both operands (both non-null; scale_block_k = 128 = the K tile): see GenericParams
    const float* sa;
    const float* sb;
    int bm, bn, ld_sa, ld_sb;
    int b_preshuf;     // weights in the pre-shuffled order (1 KiB blocks of 64 k x 32 columns of packed int4 or x 16
                       // columns of int8; column-block pitch = 32 * ldb_bytes or 16 * ldb_bytes): 256-tile kernels only
};

// Byte offset of the 16-byte unit (16 consecutive k from 16 * k16 of column n) inside pre-shuffled 8-bit weights.
__host__ __device__ __forceinline__ size_t preshuf8_unit_offset(int n, int k16, int ldb_bytes)
{
    return (size_t)(n >> 4) * ((size_t)ldb_bytes * 16) + (size_t)(k16 >> 2) * 1024 + (size_t)(((n & 15) * 4 + (k16 & 3)) * 16);
}

// Byte offset of the 16-byte unit (32 consecutive k from 32 * k32 of column n) inside pre-shuffled int4 weights.
__host__ __device__ __forceinline__ size_t preshuf_unit_offset(int n, int k32, int ldb_bytes)
{
    return (size_t)(n >> 5) * ((size_t)ldb_bytes * 32) + (size_t)(k32 >> 1) * 1024 +
           (size_t)((((n >> 4) & 1) * 32 + (k32 & 1) * 16 + (n & 15)) * 16);
}

// XCD-aware block remap: blocks b and b + 8 share an XCD (and its L2) under round-robin
// dispatch, so give each of the 8 groups a contiguous run of tile ids.  Bijective for any grid.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// Within the remapped order walk tiles in groups of 8 tile-rows so that concurrently resident
// blocks share activation rows and weight columns in L2.
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int GROUP, int& tm, int& tn)
{
    const int per_group = GROUP * tiles_n;
    const int g         = t / per_group;
    const int first_m   = g * GROUP;
    const int gsz       = min(tiles_m - first_m, GROUP);
    const int r         = t - g * per_group;
    tm                  = first_m + r % gsz;
    tn                  = r / gsz;
}


// Epilogue of an MFMA tile kernel: NT x MT accumulator fragments per wave; fragment (nt, mt) covers
// rows m_base + 16 mt + (lane & 15), columns n_base + 16 nt + 4 (lane >> 4) .. +3.  The epilogue
// kind and output type are resolved once, outside the fully unrolled fragment loops.
// RSTEP / CSTEP: row / column distance between consecutive fragments (16, 16 for 16x16 MFMA accumulators;
// 32, 8 for the register groups of 32x32 accumulators).
template <int KIND, bool BF16, int NT, int MT, int RSTEP = 16, int CSTEP = 16>
__device__ __forceinline__ void tile_epilogue_k(const GemmParams& p, const f4_t (&acc)[NT][MT], int m_base,
                                                int n_base, int fr, int fq)
{
#pragma unroll
    for(int nt = 0; nt < NT; ++nt)
#pragma unroll
        for(int mt = 0; mt < MT; ++mt)
            if(m_base + mt * RSTEP + fr < p.M && n_base + nt * CSTEP + fq * 4 < p.N) // partial last row / column tile (N % 4 == 0)
                store_c4<KIND, BF16>(p.epi, p.c, p.ldc, acc[nt][mt], m_base + mt * RSTEP + fr, n_base + nt * CSTEP + fq * 4);
}

template <int NT, int MT, int RSTEP = 16, int CSTEP = 16>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, const f4_t (&acc)[NT][MT], int m_base,
                                              int n_base, int fr, int fq)
{
    if(p.partial)
    {
        float* slab = (float*)p.c + (size_t)blockIdx.z * p.M * p.ldc;
#pragma unroll
        for(int nt = 0; nt < NT; ++nt)
#pragma unroll
            for(int mt = 0; mt < MT; ++mt)
                if(m_base + mt * RSTEP + fr < p.M && n_base + nt * CSTEP + fq * 4 < p.N)
                    *(f4_t*)(slab + (size_t)(m_base + mt * RSTEP + fr) * p.ldc + n_base + nt * CSTEP + fq * 4) = acc[nt][mt];
        return;
    }
    const bool bf = p.epi.c_dtype == 1;
    switch(p.epi.kind)
    {
    case 1:
        bf ? tile_epilogue_k<1, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<1, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 2:
        bf ? tile_epilogue_k<2, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<2, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 3:
        bf ? tile_epilogue_k<3, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<3, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 4:
        bf ? tile_epilogue_k<4, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<4, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 6:
        bf ? tile_epilogue_k<6, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<6, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 7:
        bf ? tile_epilogue_k<7, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<7, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 8:
        bf ? tile_epilogue_k<8, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<8, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 9:
        bf ? tile_epilogue_k<9, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<9, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 13: // (bf16 outputs only: checked on the host)
        tile_epilogue_k<13, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    case 14: // (bf16 outputs only)
        tile_epilogue_k<14, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    default:
        bf ? tile_epilogue_k<0, true, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq) : tile_epilogue_k<0, false, NT, MT, RSTEP, CSTEP>(p, acc, m_base, n_base, fr, fq);
        break;
    }
}

// Epilogue of the 256 x 256 tile kernels for 16-bit outputs: every wave applies the epilogue to its
// accumulators, converts and writes the 16-bit values into an LDS image of the whole C tile
// (256 rows x 512 B, row pitch 528 B), then the workgroup streams the tile out row by row with
// 16-byte stores: each wave-instruction writes two complete 512-byte rows of C, instead of the
// 16 x 32-byte pieces a direct store of an MFMA accumulator fragment makes.
#ifndef WQ_EPILOGUE_PHASES
#define WQ_EPILOGUE_PHASES 4
#endif
constexpr int kCTilePitch = 528;
constexpr int kCTileBytes = 256 * kCTilePitch;

template <int KIND, bool BF16, int NT, int MT, int RSTEP = 16, int CSTEP = 16, bool NO_STORE = false /* timing diagnostics only */>
__device__ __forceinline__ void tile256_epilogue_lds_k(const GemmParams& p, const f4_t (&acc)[NT][MT], uint8_t* lds,
                                                       int m0, int n0, int row_w, int col_w, int fr, int fq,
                                                       int wave, int lane)
{
    // When every wave's fragments span all 256 rows (1 x 8 wave grid) the tile goes out in quarters of 64 rows: the stores
    // of one quarter drain while the epilogue arithmetic of the next runs, so only the first quarter's arithmetic and the
    // last quarter's stores are exposed (halves left 128 rows of each exposed).
    constexpr int PHASES = MT * RSTEP == 256 ? WQ_EPILOGUE_PHASES : 1;
    constexpr int MTP    = MT / PHASES;
    uint16_t* cg = (uint16_t*)p.c + (size_t)m0 * p.ldc + n0;
    // D operands in their broadcast forms: one fetch per column block (and per row block for MultiplyMultiply's
    // per-row d1) instead of one per fragment
    const bool bcast = KIND != 0 && KIND != 1 && p.epi.ldd0 == 0 && p.epi.ldd1 == 0;
    f4_t d0c[NT], d1c[NT];
    if(bcast)
    {
#pragma unroll
        for(int nt = 0; nt < NT; ++nt)
        {
            const int n = n0 + col_w + nt * CSTEP + fq * 4;
            d0c[nt] = load_d0_cols4<KIND>(p.epi, n), d1c[nt] = load_d1_cols4<KIND>(p.epi, n);
        }
    }
    // (the per-row operand of every row block in one batch, ahead of the first phase: one memory latency instead of one
    // per phase)
    float d1r[MT];
    if(bcast)
    {
#pragma unroll
        for(int mt = 0; mt < MT; ++mt)
            d1r[mt] = load_d1_row<KIND>(p.epi, min(m0 + row_w + mt * RSTEP + fr, p.M - 1));
    }
#pragma unroll
    for(int ph = 0; ph < PHASES; ++ph)
    {
        if(bcast)
        {
#pragma unroll
            for(int nt = 0; nt < NT; ++nt)
#pragma unroll
                for(int mi = 0; mi < MTP; ++mi)
                {
                    const int mt = ph * MTP + mi;
                    const int r = row_w + mt * RSTEP + fr, c = col_w + nt * CSTEP + fq * 4;
                    const u2_t pk = pack_out4<BF16>(apply_epilogue4_bcast<KIND>(acc[nt][mt], d0c[nt], d1c[nt], d1r[mt]));
                    *(u2_t*)(lds + r * kCTilePitch + c * 2) = pk;
                }
        }
        else
        {
#pragma unroll
            for(int nt = 0; nt < NT; ++nt)
#pragma unroll
                for(int mi = 0; mi < MTP; ++mi)
                {
                    const int mt = ph * MTP + mi;
                    const int r = row_w + mt * RSTEP + fr, c = col_w + nt * CSTEP + fq * 4;
                    const int m = min(m0 + r, p.M - 1); // rows of a partial last tile past M: keep per-row operand reads in range
                    const u2_t pk = pack_out4<BF16>(apply_epilogue4<KIND>(p.epi, acc[nt][mt], m, n0 + c));
                    *(u2_t*)(lds + r * kCTilePitch + c * 2) = pk;
                }
        }
        __syncthreads();
        constexpr int ROWS_PH = 256 / PHASES;     // rows streamed out in this phase
        constexpr int ITERS   = ROWS_PH / 16;     // 8 waves x 2 rows per wave-instruction
#pragma unroll 4
        for(int i = 0; i < ITERS; ++i)
        {
            const int r   = ph * ROWS_PH + wave * (ROWS_PH / 8) + 2 * i + (lane >> 5);
            const int c16 = lane & 31;
            const u4_t v  = *(const u4_t*)(lds + r * kCTilePitch + c16 * 16);
            if constexpr(NO_STORE)
                asm volatile("" ::"v"(v));
            else if(m0 + r < p.M) // the last row tile may be partial
                *(u4_t*)(cg + (size_t)r * p.ldc + c16 * 8) = v;
        }
    }
}

template <int NT, int MT, int RSTEP = 16, int CSTEP = 16>
__device__ __forceinline__ void tile256_epilogue(const GemmParams& p, const f4_t (&acc)[NT][MT], uint8_t* lds, int m0,
                                                 int n0, int row_w, int col_w, int fr, int fq, int wave, int lane)
{
    if constexpr(MT * RSTEP == 256)
    {
        if(p.partial && !p.direct_store)
        {
            // fp32 partial slab through LDS, 128 rows (128 KiB + padding) at a time: every wave-instruction stores one
            // whole 1024-byte row of the slab instead of 16 pieces of 64 bytes
            constexpr int PITCH = 1024 + 16, MTP = MT / 2;
            float* slab = (float*)p.c + (size_t)blockIdx.z * p.M * p.ldc + (size_t)m0 * p.ldc + n0;
#pragma unroll
            for(int ph = 0; ph < 2; ++ph)
            {
                if(ph)
                    __syncthreads(); // the rows of the first half have left LDS
#pragma unroll
                for(int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for(int mi = 0; mi < MTP; ++mi)
                    {
                        const int mt = ph * MTP + mi;
                        const int r = row_w + mt * RSTEP + fr - ph * 128, c = col_w + nt * CSTEP + fq * 4;
                        *(f4_t*)(lds + r * PITCH + c * 4) = acc[nt][mt];
                    }
                __syncthreads();
#pragma unroll 4
                for(int i = 0; i < 16; ++i)
                {
                    const int r  = wave * 16 + i;
                    const f4_t v = *(const f4_t*)(lds + r * PITCH + lane * 16);
                    if(m0 + ph * 128 + r < p.M)
                        *(f4_t*)(slab + (size_t)(ph * 128 + r) * p.ldc + lane * 4) = v;
                }
            }
            return;
        }
    }
    if(p.partial || p.direct_store)
    {
        tile_epilogue<NT, MT, RSTEP, CSTEP>(p, acc, m0 + row_w, n0 + col_w, fr, fq);
        return;
    }
    const bool bf = p.epi.c_dtype == 1;
#define WQ_EPI_CASE(K)                                                                                              \
    case K:                                                                                                         \
        bf ? tile256_epilogue_lds_k<K, true, NT, MT, RSTEP, CSTEP>(p, acc, lds, m0, n0, row_w, col_w, fr, fq, wave, lane) \
           : tile256_epilogue_lds_k<K, false, NT, MT, RSTEP, CSTEP>(p, acc, lds, m0, n0, row_w, col_w, fr, fq, wave, lane); \
        break;
    switch(p.epi.kind)
    {
        WQ_EPI_CASE(1)
        WQ_EPI_CASE(2)
        WQ_EPI_CASE(3)
        WQ_EPI_CASE(4)
        WQ_EPI_CASE(6)
        WQ_EPI_CASE(7)
        WQ_EPI_CASE(8)
        WQ_EPI_CASE(9)
    case 13: // (bf16 outputs only: checked on the host)
        tile256_epilogue_lds_k<13, true, NT, MT, RSTEP, CSTEP>(p, acc, lds, m0, n0, row_w, col_w, fr, fq, wave, lane);
        break;
    case 14: // (bf16 outputs only)
        tile256_epilogue_lds_k<14, true, NT, MT, RSTEP, CSTEP>(p, acc, lds, m0, n0, row_w, col_w, fr, fq, wave, lane);
        break;
    default:
        bf ? tile256_epilogue_lds_k<0, true, NT, MT, RSTEP, CSTEP>(p, acc, lds, m0, n0, row_w, col_w, fr, fq, wave, lane)
           : tile256_epilogue_lds_k<0, false, NT, MT, RSTEP, CSTEP>(p, acc, lds, m0, n0, row_w, col_w, fr, fq, wave, lane);
        break;
    }
#undef WQ_EPI_CASE
}

// ---------------------------------------------------------------------------------------------
// 128 x 128 x 64 MFMA kernel (any M >= 1 and any N % 4 == 0: the last row and column tiles may be partial; K in
// whole tiles)
// ---------------------------------------------------------------------------------------------
template <int AK, int BKIND>
struct Mfma128
{
    static constexpr int BM = 128, BN = 128, BK = 64, THREADS = 256;
    static constexpr int A_ROW_BYTES = BK * 2;                                // 128
    static constexpr int B_ROW_BYTES = BK / 8 * BInfo<BKIND>::kBytesPer8K;    // 32 or 64
    static constexpr int A_TILE      = BM * A_ROW_BYTES;                      // 16 KiB
    static constexpr int B_TILE      = BN * B_ROW_BYTES;                      // 4 or 8 KiB
    static constexpr int STAGE       = A_TILE + B_TILE;
    static constexpr int LDS_BYTES   = 2 * STAGE;
};

// One 16-byte LDS-DMA per lane: LDS destination = wave-uniform base + 16 * lane.
__device__ __forceinline__ void glds16(const void* gsrc, uint8_t* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0,
                                     0);
}

// GROUPED: group-wise scales (groups of a multiple of 64 consecutive k, so one K tile lies inside one group): the
// scales of tile t + 1 are fetched while tile t is multiplied and installed behind the barrier that ends tile t.
template <int AK, int BKIND, bool GROUPED = false>
__global__ __launch_bounds__(256) void wq_gemm_mfma_128(const GemmParams p)
{
    using T = Mfma128<AK, BKIND>;
    __shared__ __attribute__((aligned(16))) uint8_t lds[T::LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr   = lane & 15; // index inside a 16-wide operand (row of the weight tile / column of the activation tile)
    const int fq   = lane >> 4; // k group: k = 8 fq .. 8 fq + 7 inside a 32-deep MFMA step

    int tm, tn;
    {
        const int nwg = p.tiles_m * p.tiles_n;
        tile_coords(xcd_remap(blockIdx.x, nwg), p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    }
    const int m0 = tm * T::BM, n0 = tn * T::BN;
    const int wm = wave & 1, wn = wave >> 1; // this wave's 64 x 64 quadrant of the block tile

    const int kz_begin = p.k_begin + blockIdx.z * p.k_per_z;
    const int kz_end   = min(p.k_end, kz_begin + p.k_per_z);
    const int ntiles   = (kz_end - kz_begin) / T::BK;

    // ---- per-lane scale registers, one per 16-column weight tile of this wave --------------
    // One coalesced load of the wave's 64 scales, then a cross-lane broadcast: lane l needs the
    // scale of column 16 nt + (l & 15).
    ScaleReg<AK> sc[4];
    // (columns past N in a partial last column tile re-read column N - 1: valid memory, results never stored)
    const int scale_col = min(n0 + wn * 64 + lane, p.N - 1);
    auto fetch_scale = [&](int k) __attribute__((always_inline)) -> uint16_t {
        if constexpr(GROUPED)
            return ((const uint16_t*)p.scale)[(size_t)(k / p.scale_group_k) * p.ld_scale + scale_col];
        else
            return p.scale ? ((const uint16_t*)p.scale)[scale_col] : unit_scale_bits<AK>();
    };
    auto install_scale = [&](uint16_t mine) __attribute__((always_inline)) {
#pragma unroll
        for(int nt = 0; nt < 4; ++nt)
            sc[nt].set((uint16_t)__shfl((int)mine, nt * 16 + fr, 64));
    };
    install_scale(fetch_scale(kz_begin));

    // ---- staging addresses -------------------------------------------------------------------
    // A tile: 128 rows x 128 B.  One wave-instruction moves 8 rows (lane -> row lane/8, 16-byte
    // chunk lane%8); the chunk fetched is XOR-swizzled with (row & 7) so that the fragment reads
    // below are bank-conflict free while the LDS image stays lane-linear.
    const uint8_t* a_src[4];
#pragma unroll
    for(int i = 0; i < 4; ++i)
    {
        const int row   = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        // rows past M (a partial last row tile) re-read row M - 1: valid memory, results never stored
        a_src[i] = (const uint8_t*)p.a + ((size_t)min(m0 + row, p.M - 1) * p.lda + kz_begin) * 2 + chunk * 16;
    }
    // B tile: 128 columns x B_ROW_BYTES.
    constexpr int B_CHUNKS = T::B_ROW_BYTES / 16;              // 16-byte chunks per column: 2 or 4
    constexpr int B_COLS_PER_INSTR = 64 / B_CHUNKS;            // 32 or 16
    constexpr int B_INSTR = T::BN / B_COLS_PER_INSTR / 4;      // per wave: 1 or 2
    const uint8_t* b_src[B_INSTR];
#pragma unroll
    for(int i = 0; i < B_INSTR; ++i)
    {
        const int col   = (wave * B_INSTR + i) * B_COLS_PER_INSTR + lane / B_CHUNKS;
        const int chunk = (lane % B_CHUNKS) ^ ((col >> 2) & (B_CHUNKS - 1));
        b_src[i]        = (const uint8_t*)p.b + (size_t)min(n0 + col, p.N - 1) * p.ldb_bytes +
                   (size_t)kz_begin / 8 * BInfo<BKIND>::kBytesPer8K + chunk * 16;
    }
    // Pre-shuffled weights (p.b_preshuf): every DMA piece is one contiguous KiB of global memory, K tile after K tile.
    // Packed int4: piece = 32 columns, stored in the order of its own LDS image (unit (column 16 t + r, k half h) at
    // slot 32 t + 16 h + r), so lane l copies slot l and only the fragment reads below change their addresses.  int8:
    // piece = 16 columns, units column after column; the lane that fills slot l fetches the unit the swizzled image
    // wants there, and the reads do not change.  Pieces past N in a partial last column tile re-read the last block.
    size_t b_tile_stride = T::B_ROW_BYTES;
    if(p.b_preshuf)
    {
        b_tile_stride = 1024;
#pragma unroll
        for(int i = 0; i < B_INSTR; ++i)
        {
            if constexpr(BKIND == B_I4P)
                b_src[i] = (const uint8_t*)p.b + (size_t)min((n0 >> 5) + wave, (p.N >> 5) - 1) * ((size_t)p.ldb_bytes * 32) +
                           (size_t)(kz_begin >> 6) * 1024 + lane * 16;
            else
            {
                const int c = lane >> 2, j = (lane & 3) ^ ((c >> 2) & 3);
                b_src[i]    = (const uint8_t*)p.b +
                           (size_t)min((n0 >> 4) + wave * B_INSTR + i, (p.N >> 4) - 1) * ((size_t)p.ldb_bytes * 16) +
                           (size_t)(kz_begin >> 6) * 1024 + (c * 4 + j) * 16;
            }
        }
    }
    // packed int4: byte offset inside the B image of this lane's word of k-step s of column block nt = 0 of its wave
    // (further column blocks: + 512 nt in both layouts)
    int raw_off[2] = {0, 0};
    if constexpr(BKIND == B_I4P)
    {
#pragma unroll
        for(int s = 0; s < 2; ++s)
            raw_off[s] = p.b_preshuf ? wn * 2048 + (s * 16 + fr) * 16 + fq * 4
                                     : (wn * 64 + fr) * 32 + ((s ^ ((fr >> 2) & 1)) << 4) + fq * 4;
    }

    auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
        uint8_t* sa = lds + buf * T::STAGE;
        uint8_t* sb = sa + T::A_TILE;
#pragma unroll
        for(int i = 0; i < 4; ++i)
            glds16(a_src[i] + (size_t)kt * T::A_ROW_BYTES, sa + (wave * 4 + i) * 1024);
#pragma unroll
        for(int i = 0; i < B_INSTR; ++i)
            glds16(b_src[i] + (size_t)kt * b_tile_stride, sb + (wave * B_INSTR + i) * 1024);
    };

    f4_t acc[4][4]; // [nt][mt]
#pragma unroll
    for(int nt = 0; nt < 4; ++nt)
#pragma unroll
        for(int mt = 0; mt < 4; ++mt)
            acc[nt][mt] = f4_t{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const uint8_t* sa = lds + buf * T::STAGE;
        const uint8_t* sb = sa + T::A_TILE;
#pragma unroll
        for(int s = 0; s < 2; ++s)
        {
            u4_t wf[4], af[4];
#pragma unroll
            for(int nt = 0; nt < 4; ++nt)
            {
                [[maybe_unused]] const int col = wn * 64 + nt * 16 + fr;
                if constexpr(BKIND == B_I4P)
                {
                    const uint32_t w = *(const uint32_t*)(sb + raw_off[s] + nt * 512);
                    wf[nt]           = dequant8(sc[nt], w, std::integral_constant<int, B_I4P>{});
                }
                else
                {
                    const int chunk = (2 * s + (fq >> 1)) ^ ((col >> 2) & 3);
                    const u2_t w    = *(const u2_t*)(sb + col * 64 + chunk * 16 + (fq & 1) * 8);
                    if constexpr(AK == A_F16)
                        wf[nt] = dequant8_bytes_f16<BKIND>(sc[nt], w);
                    else
                        wf[nt] = dequant8_bytes_bf16<BKIND>(sc[nt], w);
                }
            }
#pragma unroll
            for(int mt = 0; mt < 4; ++mt)
            {
                const int row   = wm * 64 + mt * 16 + fr;
                const int chunk = (4 * s + fq) ^ (row & 7);
                af[mt]          = *(const u4_t*)(sa + row * 128 + chunk * 16);
            }
#pragma unroll
            for(int nt = 0; nt < 4; ++nt)
#pragma unroll
                for(int mt = 0; mt < 4; ++mt)
                {
                    if constexpr(AK == A_F16)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            __builtin_bit_cast(h8_t, wf[nt]), __builtin_bit_cast(h8_t, af[mt]),
                            acc[nt][mt], 0, 0, 0);
                    else
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf8_t, wf[nt]), __builtin_bit_cast(bf8_t, af[mt]),
                            acc[nt][mt], 0, 0, 0);
                }
        }
    };

    // ---- main loop: prefetch tile t+1 by LDS-DMA while computing tile t ---------------------
    if(ntiles > 0)
    {
        stage(0, 0);
        __syncthreads(); // drains the DMA (vmcnt(0)) and publishes the tile
        for(int t = 0; t < ntiles; ++t)
        {
            const int cur = t & 1;
            uint16_t next_scale = 0;
            if(t + 1 < ntiles)
            {
                stage(cur ^ 1, t + 1);
                if constexpr(GROUPED)
                    next_scale = fetch_scale(kz_begin + (t + 1) * T::BK);
            }
            compute(cur);
            __syncthreads();
            if constexpr(GROUPED)
                if(t + 1 < ntiles)
                    install_scale(next_scale);
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------
    tile_epilogue<4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, fr, fq);
}

// Parameters of the int8 x int8 requantised GEMM kernels.
struct I8GemmParams
{
    const void* a; // int8 row-major (M, lda)
    const void* b; // int8, K contiguous per column n (ldb)
    void* c;       // int8 row-major (M, ldc)
    int M, N, K, lda, ldb, ldc;
    int tiles_m, tiles_n, group_m;
    float requant;
    int kind;            // wq_epilogue of the requantisation family (5, 10, 11, 12)
    const int32_t* bias; // N values for kinds 11 and 12
    int32_t* partial;    // non-null (128-tile kernel): blockIdx.z multiplies K tiles [z * k_tiles_per_z, ...) and stores its raw
                         // int32 accumulators to slab z of this buffer, rows of N values; the requantisation follows
                         // in wq_reduce_requant_i8
    int k_tiles_per_z;
};

#ifdef WQ_MAIN_TU // the non-template kernels below are emitted by wqgemm.hip only

// ---------------------------------------------------------------------------------------------
// int8 x int8 -> int32 MFMA kernel with the Activation_Mul_Clamp requantisation epilogue
// (example/14_gemm_quantization, client_example/09_quantization).  128 x 128 x 128 tile: with one
// byte per element a 128-deep K tile has the same 128-byte rows as the fp16 kernel's 64-deep tile, so
// staging, swizzle and fragment addressing are shared; both operands go HBM -> LDS by LDS-DMA and
// straight into v_mfma_i32_16x16x64_i8 (16 consecutive k per lane and operand).  Weights are the
// first operand: lane l owns C[m = l & 15][n = 4 (l >> 4) + j] and stores 4 int8 at once.  Any M >= 1 and any
// N % 4 == 0 (the last row and column tiles may be partial); K in whole tiles.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void wq_gemm_i8_mfma_128(const I8GemmParams p)
{
    constexpr int TILE = 128 * 128; // bytes of one operand tile
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * TILE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n), p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * 128, n0 = tn * 128;
    const int wm = wave & 1, wn = wave >> 1;
    // K tiles of this workgroup: all of them, or slice blockIdx.z (trailing slices may be empty: zero partials)
    const int kt0    = p.partial ? blockIdx.z * p.k_tiles_per_z : 0;
    const int ntiles = p.partial ? max(0, min(p.K / 128 - kt0, p.k_tiles_per_z)) : p.K / 128;

    const uint8_t *a_src[4], *b_src[4];
#pragma unroll
    for(int i = 0; i < 4; ++i)
    {
        const int row   = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        // rows past M (a partial last row tile) re-read row M - 1: valid memory, results never stored
        a_src[i]        = (const uint8_t*)p.a + (size_t)min(m0 + row, p.M - 1) * p.lda + (size_t)kt0 * 128 + chunk * 16;
        b_src[i]        = (const uint8_t*)p.b + (size_t)min(n0 + row, p.N - 1) * p.ldb + (size_t)kt0 * 128 + chunk * 16; // likewise past N
    }
    auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
        uint8_t* sa = lds + buf * 2 * TILE;
        uint8_t* sb = sa + TILE;
#pragma unroll
        for(int i = 0; i < 4; ++i)
        {
            glds16(a_src[i] + (size_t)kt * 128, sa + (wave * 4 + i) * 1024);
            glds16(b_src[i] + (size_t)kt * 128, sb + (wave * 4 + i) * 1024);
        }
    };

    i4_t acc[4][4];
#pragma unroll
    for(int nt = 0; nt < 4; ++nt)
#pragma unroll
        for(int mt = 0; mt < 4; ++mt)
            acc[nt][mt] = i4_t{0, 0, 0, 0};

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const uint8_t* sa = lds + buf * 2 * TILE;
        const uint8_t* sb = sa + TILE;
#pragma unroll
        for(int s = 0; s < 2; ++s) // two 64-deep MFMA steps per tile
        {
            i4_t wf[4], af[4];
#pragma unroll
            for(int t = 0; t < 4; ++t)
            {
                const int nrow = wn * 64 + t * 16 + fr, mrow = wm * 64 + t * 16 + fr;
                wf[t] = *(const i4_t*)(sb + nrow * 128 + (((4 * s + fq) ^ (nrow & 7)) << 4));
                af[t] = *(const i4_t*)(sa + mrow * 128 + (((4 * s + fq) ^ (mrow & 7)) << 4));
            }
#pragma unroll
            for(int nt = 0; nt < 4; ++nt)
#pragma unroll
                for(int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    };

    if(ntiles > 0)
    {
        stage(0, 0);
        __syncthreads();
        for(int t = 0; t < ntiles; ++t)
        {
            const int cur = t & 1;
            if(t + 1 < ntiles)
                stage(cur ^ 1, t + 1);
            compute(cur);
            __syncthreads();
        }
    }
    if(p.partial)
    {
        int32_t* slab = p.partial + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
        for(int nt = 0; nt < 4; ++nt)
#pragma unroll
            for(int mt = 0; mt < 4; ++mt)
            {
                const int n = n0 + wn * 64 + nt * 16 + fq * 4, m = m0 + wm * 64 + mt * 16 + fr;
                if(m < p.M && n < p.N)
                    *(i4_t*)(slab + (size_t)m * p.N + n) = acc[nt][mt];
            }
        return;
    }
    const float relu_floor = requant_relu_floor(p.kind);
#pragma unroll
    for(int nt = 0; nt < 4; ++nt)
    {
        const int n = n0 + wn * 64 + nt * 16 + fq * 4;
        i4_t bias4  = {0, 0, 0, 0};
        if(p.kind >= 11)
#pragma unroll
            for(int j = 0; j < 4; ++j)
                bias4[j] = p.bias[min(n + j, p.N - 1)];
#pragma unroll
        for(int mt = 0; mt < 4; ++mt)
        {
            const int m       = m0 + wm * 64 + mt * 16 + fr;
            const uint32_t pk = requant_pack4(acc[nt][mt], bias4, relu_floor, p.requant);
            if(m < p.M && n < p.N) // partial last row / column tile (N % 4 == 0)
                *(uint32_t*)((int8_t*)p.c + (size_t)m * p.ldc + n) = pk;
        }
    }
}


// The same GEMM for M <= 16 rows of activations (decode): one pass over the weights, bound by HBM bandwidth.
//   grid  (N / (16 NB), 1, k slices), 256 threads: the four waves of a workgroup own the same NB blocks of 16 columns and a
//         quarter of the workgroup's K range each (dealt in units of P 64-k steps; the first (units % 4) waves take one more).
//   lane  (fr, fq) streams column fr of each of its blocks: the 16 bytes at k = 64 step + 16 fq are one operand of
//         v_mfma_i32_16x16x64_i8 as they arrive, P NB loads in flight; the activations of row fr at the same k (from L2;
//         lanes of rows past M read nothing and feed rows that are never stored) are the other, fetched once for the NB
//         blocks -- at M = 16 the activation reads of NB = 1 are as many L2 requests as the weight stream itself.
//   The four partial sums meet in LDS (integer sums: exact in any order); wave nb requantises block nb and stores 4 int8
//   per lane, or stores its int32 sums to slab blockIdx.z of `partial` for wq_reduce_requant_i8.
template <int P, int NB>
__global__ __launch_bounds__(256) void wq_gemm_i8_skinny(const I8GemmParams p)
{
    __shared__ i4_t red[4 * NB * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NB);

    // K range of this workgroup (the host makes every slice a positive multiple of 64 P)
    const int kz0    = p.partial ? blockIdx.z * p.k_tiles_per_z * 128 : 0;
    const int klen   = p.partial ? min(p.K - kz0, p.k_tiles_per_z * 128) : p.K;
    const int units  = klen / (64 * P);
    const int ubase  = units >> 2, urem = units & 3;
    const int nsteps = (ubase + (wave < urem ? 1 : 0)) * P;
    // (a wave without work still issues its first loads: it points them at the start of the range)
    const int kbeg   = kz0 + (nsteps ? (wave * ubase + min(wave, urem)) * (64 * P) : 0);

    const uint8_t* bptr[NB];
#pragma unroll
    for(int nb = 0; nb < NB; ++nb)
        bptr[nb] = (const uint8_t*)p.b + (size_t)(n0 + nb * 16 + fr) * p.ldb + kbeg + fq * 16;
    // activations through a buffer resource that ends with row M - 1: lanes of the rows past it carry an offset
    // beyond the end, which reads as zero without touching memory
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)((uint32_t)(p.M - 1) * (uint32_t)p.lda + (uint32_t)p.K), 0x00020000);
    const uint32_t a_voff = fr < p.M ? (uint32_t)fr * (uint32_t)p.lda + (uint32_t)(kbeg + fq * 16) : 0x7F000000u;

    i4_t wraw[NB][P], araw[P];
#pragma unroll
    for(int i = 0; i < P; ++i)
    {
        const int st = min(i, max(nsteps - 1, 0));
#pragma unroll
        for(int nb = 0; nb < NB; ++nb)
            wraw[nb][i] = *(const i4_t*)(bptr[nb] + (size_t)st * 64);
        araw[i] = __builtin_bit_cast(i4_t, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff + st * 64, 0, 0));
    }
    i4_t acc[NB];
#pragma unroll
    for(int nb = 0; nb < NB; ++nb)
        acc[nb] = i4_t{0, 0, 0, 0};
    for(int t = 0; t < nsteps; t += P)
    {
#pragma unroll
        for(int i = 0; i < P; ++i)
        {
            const i4_t a  = araw[i];
            const int nxt = min(t + i + P, nsteps - 1);
            i4_t w[NB];
#pragma unroll
            for(int nb = 0; nb < NB; ++nb)
            {
                w[nb]       = wraw[nb][i];
                wraw[nb][i] = *(const i4_t*)(bptr[nb] + (size_t)nxt * 64);
            }
            araw[i] = __builtin_bit_cast(i4_t, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff + nxt * 64, 0, 0));
#pragma unroll
            for(int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[nb], a, acc[nb], 0, 0, 0);
        }
    }

#pragma unroll
    for(int nb = 0; nb < NB; ++nb)
        red[(wave * NB + nb) * 64 + lane] = acc[nb];
    __syncthreads();
    if(wave >= NB || fr >= p.M)
        return;
    i4_t v = red[wave * 64 + lane]; // block `wave` of wave 0, then of waves 1 .. 3
#pragma unroll
    for(int w = 1; w < 4; ++w)
    {
        const i4_t u = red[(w * NB + wave) * 64 + lane];
        v[0] += u[0], v[1] += u[1], v[2] += u[2], v[3] += u[3];
    }
    const int m = fr, n = n0 + wave * 16 + fq * 4;
    if(p.partial)
    {
        *(i4_t*)(p.partial + ((size_t)blockIdx.z * p.M + m) * p.N + n) = v;
        return;
    }
    i4_t bias4 = {0, 0, 0, 0};
    if(p.kind >= 11)
#pragma unroll
        for(int j = 0; j < 4; ++j)
            bias4[j] = p.bias[n + j];
    *(uint32_t*)((int8_t*)p.c + (size_t)m * p.ldc + n) = requant_pack4(v, bias4, requant_relu_floor(p.kind), p.requant);
}

// The same GEMM on 256 x 256 x 128 tiles for problems that fill the chip: 8 waves in a 2 x 4 grid (each 128 rows x 64
// columns = 8 x 4 accumulator fragments), both operands by LDS-DMA into a two-stage ring (2 x 64 KiB), 24 fragment
// reads for 64 MFMAs per K tile and wave (the 128-tile kernel: 16 for 32), and the requantised int8 tile staged
// through LDS so that it leaves as whole 256-byte rows.  Any M >= 1 (partial last row tile).
constexpr int kI8Tile256Lds = 4 * 256 * 128;

__global__ __launch_bounds__(512) void wq_gemm_i8_mfma_256(const I8GemmParams p)
{
    constexpr int TILE = 256 * 128; // bytes of one operand tile
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n), p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const int wm = wave & 1, wn = wave >> 1;
    const int ntiles = p.K / 128;

    const uint8_t *a_src[4], *b_src[4];
#pragma unroll
    for(int i = 0; i < 4; ++i)
    {
        const int row   = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        a_src[i]        = (const uint8_t*)p.a + (size_t)min(m0 + row, p.M - 1) * p.lda + chunk * 16;
        b_src[i]        = (const uint8_t*)p.b + (size_t)(n0 + row) * p.ldb + chunk * 16;
    }
    auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
        uint8_t* sa = lds + buf * 2 * TILE;
        uint8_t* sb = sa + TILE;
#pragma unroll
        for(int i = 0; i < 4; ++i)
        {
            glds16(a_src[i] + (size_t)kt * 128, sa + (wave * 4 + i) * 1024);
            glds16(b_src[i] + (size_t)kt * 128, sb + (wave * 4 + i) * 1024);
        }
    };

    i4_t acc[4][8]; // [nt][mt]
#pragma unroll
    for(int nt = 0; nt < 4; ++nt)
#pragma unroll
        for(int mt = 0; mt < 8; ++mt)
            acc[nt][mt] = i4_t{0, 0, 0, 0};

    // Ping-pong: waves 0-3 and waves 4-7 (one of each per SIMD) run the same sequence of sections half a step apart,
    // so that while one group issues its 32 MFMAs of a 64-deep step the other fetches the 12 fragments of its next
    // step (and, once per K tile, issues its share of the LDS-DMA of tile t + 1).  Section ends are workgroup barriers;
    // the late group passes one extra barrier first and the early group one extra at the end.
    //   L(t, s): [s == 0: DMA of tile t + 1 into the other stage]  fragment reads of step s   [s == 1: own DMA landed]
    //   M(t, s): 32 MFMAs
    // The stage tile t + 1 goes to was last read in L(t - 1, 1) of both groups, which lies behind a barrier before
    // either group reaches L(t, 0); tile t + 1 is first read in L(t + 1, 0), behind the barrier that ends L(t, 1) of
    // both groups, where every wave has waited for its own DMA.
    const int late = wave >> 2;
    i4_t wf[4], af[8];
    auto load_frags = [&](int buf, int s) __attribute__((always_inline)) {
        const uint8_t* sa = lds + buf * 2 * TILE;
        const uint8_t* sb = sa + TILE;
#pragma unroll
        for(int t = 0; t < 4; ++t)
        {
            const int nrow = wn * 64 + t * 16 + fr;
            wf[t]          = *(const i4_t*)(sb + nrow * 128 + (((4 * s + fq) ^ (nrow & 7)) << 4));
        }
#pragma unroll
        for(int t = 0; t < 8; ++t)
        {
            const int mrow = wm * 128 + t * 16 + fr;
            af[t]          = *(const i4_t*)(sa + mrow * 128 + (((4 * s + fq) ^ (mrow & 7)) << 4));
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for(int mt = 0; mt < 8; ++mt)
#pragma unroll
            for(int nt = 0; nt < 4; ++nt)
                acc[nt][mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        // MFMAs have no side effects the optimiser knows of: an opaque use keeps them inside their section
#pragma unroll
        for(int mt = 0; mt < 8; ++mt)
            asm volatile("" : "+v"(acc[0][mt]), "+v"(acc[1][mt]), "+v"(acc[2][mt]), "+v"(acc[3][mt]));
        __builtin_amdgcn_s_setprio(0);
    };
    auto section_end = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if(ntiles > 0)
    {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if(late)
            __builtin_amdgcn_s_barrier();
        for(int t = 0; t < ntiles; ++t)
        {
            const int cur = t & 1;
            // L(t, 0)
            if(t + 1 < ntiles)
                stage(cur ^ 1, t + 1);
            load_frags(cur, 0);
            section_end();
            // M(t, 0)
            mfmas();
            section_end();
            // L(t, 1)
            load_frags(cur, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            section_end();
            // M(t, 1)
            mfmas();
            section_end();
        }
        if(!late)
            __builtin_amdgcn_s_barrier();
    }
    __syncthreads();
    // requantise into an LDS image of the tile (256 rows x 256 bytes, pitch 272), then stream whole rows out
    constexpr int PITCH = 256 + 16;
    const float relu_floor = requant_relu_floor(p.kind);
#pragma unroll
    for(int nt = 0; nt < 4; ++nt)
    {
        const int c = wn * 64 + nt * 16 + fq * 4;
        i4_t bias4  = {0, 0, 0, 0};
        if(p.kind >= 11)
#pragma unroll
            for(int j = 0; j < 4; ++j)
                bias4[j] = p.bias[n0 + c + j];
#pragma unroll
        for(int mt = 0; mt < 8; ++mt)
        {
            const int r                        = wm * 128 + mt * 16 + fr;
            *(uint32_t*)(lds + r * PITCH + c) = requant_pack4(acc[nt][mt], bias4, relu_floor, p.requant);
        }
    }
    __syncthreads();
    int8_t* cg = (int8_t*)p.c + (size_t)m0 * p.ldc + n0;
#pragma unroll
    for(int i = 0; i < 8; ++i)
    {
        const int r   = wave * 32 + 4 * i + (lane >> 4);
        const int c16 = lane & 15;
        const u4_t v  = *(const u4_t*)(lds + r * PITCH + c16 * 16);
        if(m0 + r < p.M)
            *(u4_t*)(cg + (size_t)r * p.ldc + c16 * 16) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Generic kernel: any M, N, K slice and strides; one thread per output element, fp32 accumulate
// in ascending k (the reference's own evaluation order).
// ---------------------------------------------------------------------------------------------
struct GenericParams
{
    const void* a;
    const void* b;
    const void* scale;
    void* c;
    int M, N, K;
    long a_sm, a_sk; // element strides of A(m,k)
    long b_sk, b_sn; // element strides of B(k,n); for packed int4: b_sn in BYTES, K contiguous
    int ldc;
    int k_begin, k_end, k_per_z, partial;
    int a_dtype, b_dtype; // wq_dtype
    EpilogueArgs epi;
    int scale_group_k, ld_scale; // > 0: group-wise scales, scale[(k / scale_group_k) * ld_scale + n]
    // fp8 activations with block-wise scales of both operands (both non-null): a' = float(a) * sa[(m / bm) * ld_sa + k / bk],
    // b' = float(q) * sb[k / bk + (n / bn) * ld_sb]
    const float* sa;
    const float* sb;
    int bm, bn, bk, ld_sa, ld_sb;
};

// 8-bit weights (K, N) row-major (N contiguous, ld = ldb) -> (N, K) with K contiguous, the operand order of the
// matrix-core kernels.  64 x 64 byte tiles through LDS; K and N are multiples of 64.
__global__ __launch_bounds__(256) void wq_transpose_b8(const uint8_t* b, int ldb, int K, int N, uint8_t* out)
{
    __shared__ uint8_t tile[64][64 + 4];
    (void)N; // the grid (N / 64, K / 64) carries the extent
    const int k0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int t  = threadIdx.x;
    // read: thread t -> row k = t / 4 (+0), 16 bytes at n = 16 (t % 4)
    {
        const int k = t >> 2, nq = (t & 3) * 16;
        const u4_t v = *(const u4_t*)(b + (size_t)(k0 + k) * ldb + n0 + nq);
#pragma unroll
        for(int i = 0; i < 4; ++i)
            *(uint32_t*)&tile[k][nq + 4 * i] = v[i];
    }
    __syncthreads();
    // write: thread t -> column n = t / 4, 16 consecutive k starting at 16 (t % 4)
    {
        const int n = t >> 2, kq = (t & 3) * 16;
        u4_t v;
#pragma unroll
        for(int i = 0; i < 4; ++i)
        {
            uint32_t w = 0;
#pragma unroll
            for(int j = 0; j < 4; ++j)
                w |= (uint32_t)tile[kq + 4 * i + j][n] << (8 * j);
            v[i] = w;
        }
        *(u4_t*)(out + (size_t)(n0 + n) * K + k0 + kq) = v;
    }
}

// fp8 activations (FNUZ, or OCP E4M3FN when fn != 0) -> fp16 (exact: every value of either format, NaN included, is
// an fp16 value), 8 per thread, into a dense [M][K] image; also fills `ones` (N fp16 values 1.0, the unit weight scale
// of the fp16 GEMM that follows).
__global__ __launch_bounds__(256) void wq_f8_fnuz_to_f16(const uint8_t* a, int lda, int M, int K, uint16_t* out,
                                                         uint16_t* ones, int N, int fn)
{
    const long groups = (long)M * (K / 8);
    for(long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long)gridDim.x * blockDim.x)
    {
        const int m = (int)(g / (K / 8)), k8 = (int)(g % (K / 8));
        const u2_t raw = *(const u2_t*)(a + (size_t)m * lda + k8 * 8);
        u4_t o;
#pragma unroll
        for(int i = 0; i < 4; ++i)
        {
            const uint32_t w  = raw[i >> 1] >> (16 * (i & 1));
            const uint8_t b0 = (uint8_t)(w & 0xFF), b1 = (uint8_t)((w >> 8) & 0xFF);
            const _Float16 lo = (_Float16)(fn ? f8_e4m3fn_to_f32(b0) : f8_fnuz_to_f32(b0));
            const _Float16 hi = (_Float16)(fn ? f8_e4m3fn_to_f32(b1) : f8_fnuz_to_f32(b1));
            o[i] = (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
        }
        *(u4_t*)(out + (size_t)m * K + k8 * 8) = o;
    }
    for(int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x)
        ones[n] = 0x3C00;
}

__global__ __launch_bounds__(256) void wq_gemm_generic(const GenericParams p)
{
    const int n = blockIdx.x * 16 + (threadIdx.x & 15);
    const int m = blockIdx.y * 16 + (threadIdx.x >> 4);
    if(m >= p.M || n >= p.N)
        return;
    const int kb = p.k_begin + blockIdx.z * p.k_per_z;
    const int ke = min(p.k_end, kb + p.k_per_z);

    if(p.a_dtype == 4 /*WQ_I8*/)
    {
        // int8 x int8 -> int32 (exact), then the requantisation functor
        int acc = 0;
        for(int k = kb; k < ke; ++k)
            acc += (int)((const int8_t*)p.a)[m * p.a_sm + k * p.a_sk] *
                   (int)((const int8_t*)p.b)[k * p.b_sk + n * p.b_sn];
        ((int8_t*)p.c)[(size_t)m * p.ldc + n] =
            requant_epilogue(p.epi.kind, p.epi.requant, acc, (const int32_t*)p.epi.d0, n);
        return;
    }

    uint16_t sbits = p.scale ? ((const uint16_t*)p.scale)[n] : (p.a_dtype == 1 /*WQ_BF16*/ ? 0x3F80 : 0x3C00);
    float acc      = 0.f;
    for(int k = kb; k < ke; ++k)
    {
        if(p.scale_group_k > 0)
            sbits = ((const uint16_t*)p.scale)[(size_t)(k / p.scale_group_k) * p.ld_scale + n];
        int q;
        if(p.b_dtype == 6 /*WQ_I4_PACKED*/ || p.b_dtype == 9 /*WQ_I4_PRESHUFFLED*/)
        {
            const uint8_t* unit = p.b_dtype == 6 ? (const uint8_t*)p.b + (size_t)n * p.b_sn + (k >> 5) * 16
                                                 : (const uint8_t*)p.b + preshuf_unit_offset(n, k >> 5, p.b_sn);
            const uint32_t w = *(const uint32_t*)(unit + ((k >> 3) & 3) * 4);
            const int j      = k & 7;
            const int pos    = (j & 1) ? 4 + (j >> 1) : (j >> 1);
            q                = (int)((w >> (4 * pos)) & 0xFu) - 8;
        }
        else if(p.b_dtype == 10 /*WQ_I8_PRESHUFFLED*/)
            q = (int)((const int8_t*)p.b)[preshuf8_unit_offset(n, k >> 4, (int)p.b_sn) + (k & 15)];
        else if(p.b_dtype == 5 /*WQ_U8_OFFSET*/)
            q = (int)((const uint8_t*)p.b)[k * p.b_sk + n * p.b_sn] - 128;
        else
            q = (int)((const int8_t*)p.b)[k * p.b_sk + n * p.b_sn];

        float va, vb;
        if(p.a_dtype == 0 /*WQ_F16*/)
        {
            va = (float)((const _Float16*)p.a)[m * p.a_sm + k * p.a_sk];
            vb = dequant_scalar<A_F16>(q, sbits);
        }
        else if(p.a_dtype == 1 /*WQ_BF16*/)
        {
            va = bf16_to_f32(((const uint16_t*)p.a)[m * p.a_sm + k * p.a_sk]);
            vb = dequant_scalar<A_BF16>(q, sbits);
        }
        else /* fp8: scales live in the epilogue */
        {
            const uint8_t x = ((const uint8_t*)p.a)[m * p.a_sm + k * p.a_sk];
            va              = p.a_dtype == 8 /*WQ_F8_E4M3FN*/ ? f8_e4m3fn_to_f32(x) : f8_fnuz_to_f32(x);
            vb              = (float)q;
            if(p.sa)
            {
                // example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:262-283: both operands are
                // dequantised to fp32 (one rounding each) before the fp32 multiply-add
                va = va * p.sa[(size_t)(m / p.bm) * p.ld_sa + k / p.bk];
                vb = vb * p.sb[(size_t)(k / p.bk) + (size_t)(n / p.bn) * p.ld_sb];
            }
        }
        acc += va * vb;
    }
    if(p.partial)
        ((float*)p.c)[(size_t)blockIdx.z * p.M * p.ldc + (size_t)m * p.ldc + n] = acc;
    else
        ((uint16_t*)p.c)[(size_t)m * p.ldc + n] =
            cvt_out16(apply_epilogue(p.epi, acc, m, n), p.epi.c_dtype);
}

// Split-K of the int8 x int8 GEMM: sum the int32 slabs (exact in any order), then the requantisation of the kernels above.
__global__ __launch_bounds__(256) void wq_reduce_requant_i8(const int32_t* __restrict__ partials, int k_batch, int M, int N,
                                                            int8_t* c, int ldc, int kind, float requant,
                                                            const int32_t* __restrict__ bias)
{
    const int n4           = N >> 2; // N % 4 == 0 is checked on the host
    const long total       = (long)M * n4;
    const float relu_floor = requant_relu_floor(kind);
    for(long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    {
        const int m = (int)(i / n4);
        const int n = (int)(i % n4) * 4;
        i4_t s      = *(const i4_t*)(partials + (size_t)m * N + n);
        for(int z = 1; z < k_batch; ++z)
        {
            const i4_t v = *(const i4_t*)(partials + ((size_t)z * M + m) * N + n);
            s[0] += v[0], s[1] += v[1], s[2] += v[2], s[3] += v[3];
        }
        i4_t bias4 = {0, 0, 0, 0};
        if(kind >= 11)
#pragma unroll
            for(int j = 0; j < 4; ++j)
                bias4[j] = bias[n + j];
        *(uint32_t*)(c + (size_t)m * ldc + n) = requant_pack4(s, bias4, relu_floor, requant);
    }
}

// ---------------------------------------------------------------------------------------------
// Split-K reduce + epilogue: out(m,n) = cvt(epilogue(sum_z partial[z][m][n])), z ascending.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wq_reduce_finalize(const float* __restrict__ partials,
                                                          int k_batch, int M, int N, int ld_part,
                                                          void* c, int ldc, EpilogueArgs epi)
{
    const int n4    = N >> 2; // N % 4 == 0 is checked on the host
    const long total = (long)M * n4;
    for(long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
        i += (long)gridDim.x * blockDim.x)
    {
        const int m = (int)(i / n4);
        const int n = (int)(i % n4) * 4;
        f4_t s      = *(const f4_t*)(partials + (size_t)m * ld_part + n);
        for(int z = 1; z < k_batch; ++z)
        {
            const f4_t v = *(const f4_t*)(partials + ((size_t)z * M + m) * ld_part + n);
            s[0] += v[0];
            s[1] += v[1];
            s[2] += v[2];
            s[3] += v[3];
        }
        if(epi.c_dtype == 3 /*WQ_F32*/)
        {
            f4_t o;
#pragma unroll
            for(int j = 0; j < 4; ++j)
                o[j] = apply_epilogue(epi, s[j], m, n + j);
            *(f4_t*)((float*)c + (size_t)m * ldc + n) = o;
        }
        else
        {
            uint16_t o[4];
#pragma unroll
            for(int j = 0; j < 4; ++j)
                o[j] = cvt_out16(apply_epilogue(epi, s[j], m, n + j), epi.c_dtype);
            u2_t pk;
            pk[0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            pk[1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *(u2_t*)((uint16_t*)c + (size_t)m * ldc + n) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// int4 packer: one thread per 32-bit word (8 consecutive k of one column).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wq_pack_int4(const int8_t* __restrict__ q, int ldq, int K,
                                                    int N, uint8_t* __restrict__ out, int ld_bytes)
{
    const int words = K >> 3;
    const long total = (long)N * words;
    for(long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
        i += (long)gridDim.x * blockDim.x)
    {
        const int n = (int)(i / words), g = (int)(i % words);
        const int8_t* src = q + (size_t)n * ldq + 8 * g;
        uint32_t w        = 0;
#pragma unroll
        for(int pos = 0; pos < 8; ++pos)
        {
            const int k = pos < 4 ? 2 * pos : 2 * (pos - 4) + 1;
            w |= (uint32_t)((src[k] + 8) & 0xF) << (4 * pos);
        }
        *(uint32_t*)(out + (size_t)n * ld_bytes + 4 * g) = w;
    }
}

// ---------------------------------------------------------------------------------------------
// packed int4 -> pre-shuffled int4: one thread per 16-byte unit (32 consecutive k of one column).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wq_preshuffle_int4(const uint8_t* __restrict__ packed, int ld_bytes, int K, int N,
                                                          uint8_t* __restrict__ out, int ldb_bytes_out)
{
    const int units  = K >> 5;
    const long total = (long)N * units;
    for(long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    {
        // consecutive threads write consecutive units of the destination
        const long blk = i >> 6;
        const int u    = (int)(i & 63);
        const int kbs  = K >> 6; // 64-k blocks per column block
        const int cb = (int)(blk / kbs), kb = (int)(blk % kbs);
        const int n = cb * 32 + (u >> 5) * 16 + (u & 15), k32 = kb * 2 + ((u >> 4) & 1);
        const u4_t v = *(const u4_t*)(packed + (size_t)n * ld_bytes + (size_t)k32 * 16);
        *(u4_t*)(out + preshuf_unit_offset(n, k32, ldb_bytes_out)) = v;
    }
}

// int8 (K contiguous per column) -> pre-shuffled int8: one thread per 16-byte unit (16 consecutive k of one column).
__global__ __launch_bounds__(256) void wq_preshuffle_int8(const uint8_t* __restrict__ b, int ldb, int K, int N,
                                                          uint8_t* __restrict__ out, int ldb_out)
{
    const int units  = K >> 4;
    const long total = (long)N * units;
    for(long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    {
        const long blk = i >> 6;
        const int u    = (int)(i & 63);
        const int kbs  = K >> 6;
        const int cb = (int)(blk / kbs), kb = (int)(blk % kbs);
        const int n = cb * 16 + (u >> 2), k16 = kb * 4 + (u & 3);
        const u4_t v = *(const u4_t*)(b + (size_t)n * ldb + (size_t)k16 * 16);
        *(u4_t*)(out + preshuf8_unit_offset(n, k16, ldb_out)) = v;
    }
}

#endif // WQ_MAIN_TU

} // namespace wq
