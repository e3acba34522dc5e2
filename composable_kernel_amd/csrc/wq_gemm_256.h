// 256 x 256 x 64 MFMA kernel of the weight-quantized GEMM engine (gfx950).
//
// One 512-thread workgroup (8 waves = 2 per SIMD) per 256 x 256 tile of C, one workgroup per CU.
// Waves form a WM (M) x 8/WM (N) grid:
//   WM = 2: wave (w >> 2, w & 3) owns 128 rows x 64 columns (8 x 4 accumulator tiles of
//           v_mfma_f32_16x16x32); SIMD partners (waves w, w + 4) share their weight columns, so every
//           weight is dequantised twice per workgroup;
//   WM = 1: wave w owns all 256 rows x 32 columns (16 x 2 accumulator tiles): every weight is
//           dequantised exactly once per workgroup (half the VALU work per MFMA) at the price of
//           twice the activation-fragment LDS reads.
//
// Staging: A (256 rows x 128 B) and the packed weights (256 columns x 32 or 64 B) of one K tile
// are moved HBM -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of NBUF = 3 buffers, two K
// tiles ahead of the MFMAs.  A wave waits for its own DMA of the NEXT tile with a counted
// s_waitcnt vmcnt(N) that leaves the tile after it in flight, then all waves meet at ONE raw
// s_barrier per K tile.  That barrier is both the "tile t+1 has landed for everybody" point (read
// a staged buffer only after the wait that retires it AND a barrier) and the "everybody is done
// reading tile t" point (its buffer is re-staged only in the following iteration).
//
// The weight fragment is the first MFMA operand, the activation fragment the second, so lane l
// holds C[m = l & 15][n = 4 (l >> 4) + j]: four consecutive n of a row-major C per accumulator.
#pragma once

#include "wq_kernels.h"

namespace wq {

template <int BKIND>
struct Mfma256
{
    static constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512, NBUF = 3;
    static constexpr int A_ROW_BYTES = BK * 2;                              // 128
    static constexpr int B_ROW_BYTES = BK / 8 * BInfo<BKIND>::kBytesPer8K;  // 32 or 64
    static constexpr int A_TILE      = BM * A_ROW_BYTES;                    // 32 KiB
    static constexpr int B_TILE      = BN * B_ROW_BYTES;                    // 8 or 16 KiB
    static constexpr int STAGE       = A_TILE + B_TILE;
    static constexpr int RING_BYTES  = NBUF * STAGE;                        // 120 or 144 KiB
    static constexpr int LDS_BYTES   = RING_BYTES > kCTileBytes ? RING_BYTES : kCTileBytes; // the C tile image reuses the ring
    static constexpr int A_DMA       = A_TILE / 1024 / 8;                   // LDS-DMA instructions per wave per tile: 4
    static constexpr int B_DMA       = B_TILE / 1024 / 8;                   // 1 or 2
    static constexpr int DMA_PER_TILE = A_DMA + B_DMA;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// (w & mask) | magic in one VALU instruction; mask and magic live in registers.
__device__ __forceinline__ uint32_t and_or(uint32_t w, uint32_t mask, uint32_t magic)
{
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(w), "v"(mask), "v"(magic));
    return r;
}

struct I4Consts
{
    uint32_t mask_lo, mask_hi, magic;
    h2_t k1032, k16th, km72;
    __device__ __forceinline__ void init()
    {
        mask_lo = 0x000F000Fu;
        mask_hi = 0x00F000F0u;
        magic   = 0x64006400u;
        // keep the three words in VGPRs so v_and_or_b32 can take them (one literal at most otherwise)
        asm volatile("" : "+v"(mask_lo), "+v"(mask_hi), "+v"(magic));
        k1032 = h2_t{(_Float16)1032.0f, (_Float16)1032.0f};
        k16th = h2_t{(_Float16)0.0625f, (_Float16)0.0625f};
        km72  = h2_t{(_Float16)-72.0f, (_Float16)-72.0f};
    }
};

// fp16 x packed int4, 13 VALU per 8 weights (see dequant8 in wq_device.h for the arithmetic).
__device__ __forceinline__ u4_t dequant8_i4_f16(const ScaleReg<A_F16>& sc, const I4Consts& c, uint32_t w)
{
    const uint32_t w8 = w >> 8;
    const h2_t t0     = as_h2(and_or(w, c.mask_lo, c.magic));
    const h2_t t1     = as_h2(and_or(w, c.mask_hi, c.magic));
    const h2_t t2     = as_h2(and_or(w8, c.mask_lo, c.magic));
    const h2_t t3     = as_h2(and_or(w8, c.mask_hi, c.magic));
    u4_t r;
    r[0] = as_u32((t0 - c.k1032) * sc.s2);
    r[1] = as_u32(__builtin_elementwise_fma(t1, c.k16th, c.km72) * sc.s2);
    r[2] = as_u32((t2 - c.k1032) * sc.s2);
    r[3] = as_u32(__builtin_elementwise_fma(t3, c.k16th, c.km72) * sc.s2);
    return r;
}

// SCHED: 0 lockstep (one barrier per K tile),
//        2 ping-pong, dequantisation in the L sections, DMA of tile t+2 issued between the MFMAs of M1(t),
//        5 ping-pong where the L sections only fetch activation fragments (plus the DMA issue) and each
//          MFMA section also dequantises, between its MFMAs, the weight fragments of its NEXT section.
// The fp16 x int4 dequantisation cut into three short stages (bit tricks / exact integer / scale) so
// that an MFMA section can slot a few VALU instructions between groups of MFMAs.  Applied in place.
template <int STAGE>
__device__ __forceinline__ void dequant8_i4_f16_stage(u4_t& v, const ScaleReg<A_F16>& sc, const I4Consts& c, uint32_t w)
{
    if constexpr(STAGE == 0)
    {
        const uint32_t w8 = w >> 8;
        v[0]              = and_or(w, c.mask_lo, c.magic);
        v[1]              = and_or(w, c.mask_hi, c.magic);
        v[2]              = and_or(w8, c.mask_lo, c.magic);
        v[3]              = and_or(w8, c.mask_hi, c.magic);
    }
    else if constexpr(STAGE == 1)
    {
        v[0] = as_u32(as_h2(v[0]) - c.k1032);
        v[1] = as_u32(__builtin_elementwise_fma(as_h2(v[1]), c.k16th, c.km72));
        v[2] = as_u32(as_h2(v[2]) - c.k1032);
        v[3] = as_u32(__builtin_elementwise_fma(as_h2(v[3]), c.k16th, c.km72));
    }
    else
    {
#pragma unroll
        for(int i = 0; i < 4; ++i)
            v[i] = as_u32(as_h2(v[i]) * sc.s2);
    }
}

// STAMP: diagnostic build only -- every wave of workgroup 0 accumulates, per section kind, the shader
// cycles spent working and the cycles spent waiting at the section's barrier, and writes the sums to
// p.dbg (no output value depends on them; the product never launches this instantiation).
//
// GROUPED (SCHED 5): group-wise weight scales, groups of a multiple of 64 consecutive k, so a K
// tile lies inside one group.  The scales of a wave's own columns travel with the tile: one more LDS-DMA piece
// (4 bytes per lane) behind the first part of tile t+2's pieces in L0(t), into a private 256-byte LDS area per wave and
// tile parity behind the ring; the counted vmcnt waits allow one more operation in flight.  The scale registers are
// swapped at the head of M1(t), between the dequantisation of (t, k-step 1) and that of (t+1, k-step 0).
constexpr int kGroupScaleLds = 8 * 2 * 256; // bytes of dynamic LDS behind Mfma256<>::LDS_BYTES in the GROUPED instances

template <int AK, int BKIND, int WM, int SCHED, bool STAMP = false, bool GROUPED = false>
__global__ __launch_bounds__(512, 2) void wq_gemm_mfma_256(const GemmParams p)
{
    static_assert(!GROUPED || SCHED == 5, "group-wise scales: default schedule");
    using T = Mfma256<BKIND>;
    constexpr int WN = 8 / WM;        // waves along N
    constexpr int MT = 256 / WM / 16; // accumulator tiles along M per wave: 8 or 16
    constexpr int NT = 256 / WN / 16; // accumulator tiles along N per wave: 4 or 2
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int lane  = threadIdx.x & 63;
    const int wave  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr    = lane & 15;
    const int fq    = lane >> 4;
    const int wm    = WM == 2 ? wave >> 2 : 0;
    const int wn    = WM == 2 ? wave & 3 : wave;
    const int late  = wave >> 2; // second wave of its SIMD: runs one barrier behind in the ping-pong schedules
    const int row_w = wm * (MT * 16);
    const int col_w = wn * (NT * 16);

    int tm, tn;
    {
        const int nwg = p.tiles_m * p.tiles_n;
        tile_coords(xcd_remap(blockIdx.x, nwg), p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    }
    const int m0 = tm * T::BM, n0 = tn * T::BN;

    const int kz_begin = p.k_begin + blockIdx.z * p.k_per_z;
    const int kz_end   = min(p.k_end, kz_begin + p.k_per_z);
    const int ntiles   = (kz_end - kz_begin) / T::BK;

    // per-channel scales: one coalesced load per wave, then a cross-lane broadcast (lane l needs
    // the scale of column 16 nt + (l & 15) of the wave's NT * 16 columns)
    ScaleReg<AK> sc[NT];
    {
        uint16_t mine;
        if constexpr(GROUPED)
            mine = ((const uint16_t*)p.scale)[(size_t)(kz_begin / p.scale_group_k) * p.ld_scale + n0 + col_w + (lane & (NT * 16 - 1))];
        else
            mine = p.scale ? ((const uint16_t*)p.scale)[n0 + col_w + (lane & (NT * 16 - 1))] : unit_scale_bits<AK>();
#pragma unroll
        for(int nt = 0; nt < NT; ++nt)
            sc[nt].set((uint16_t)__shfl((int)mine, nt * 16 + fr, 64));
    }
    // GROUPED: this wave's scale area and the per-lane source of its LDS-DMA (lane l moves the scales of columns
    // col_w + 2 l, + 1; lanes past the wave's NT * 16 columns move neighbours' scales nobody reads, clamped into the row)
    uint8_t* const sc_lds = lds + T::LDS_BYTES + wave * 512;
    const uint8_t* sc_src = (const uint8_t*)p.scale + (size_t)min(n0 + col_w + 2 * lane, p.N - 2) * 2;
    int sc_rem            = 0; // k offset of the last tile handed to dma_scale inside its group
    if constexpr(GROUPED)
    {
        sc_src += (size_t)(kz_begin / p.scale_group_k) * p.ld_scale * 2;
        sc_rem = kz_begin % p.scale_group_k;
    }
    // called once per K tile in ascending order, from tile 1 on
    auto dma_scale = [&](int tile) __attribute__((always_inline)) {
        sc_rem += T::BK;
        if(sc_rem >= p.scale_group_k)
        {
            sc_rem -= p.scale_group_k;
            sc_src += (size_t)p.ld_scale * 2;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sc_src,
                                         (__attribute__((address_space(3))) void*)(sc_lds + (tile & 1) * 256), 4, 0, 0);
    };
    uint16_t sc_next[NT];
    I4Consts i4c;
    i4c.init();

    // ---- LDS-DMA source addresses (per lane) and destinations (per wave) ---------------------
    // A: wave-instruction i of wave w moves rows 8 (4 w + i) .. +7; lane -> row lane / 8, 16-byte
    // position lane % 8, which receives global chunk position ^ (row & 7).
    const uint8_t* a_src[T::A_DMA];
#pragma unroll
    for(int i = 0; i < T::A_DMA; ++i)
    {
        const int row   = (wave * T::A_DMA + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        // rows past M (a partial last row tile) re-read row M - 1: valid memory, results never stored
        a_src[i] = (const uint8_t*)p.a + ((size_t)min(m0 + row, p.M - 1) * p.lda + kz_begin) * 2 + chunk * 16;
    }
    constexpr int B_CHUNKS         = T::B_ROW_BYTES / 16; // 2 or 4
    constexpr int B_COLS_PER_INSTR = 64 / B_CHUNKS;       // 32 or 16
    const uint8_t* b_src[T::B_DMA];
#pragma unroll
    for(int i = 0; i < T::B_DMA; ++i)
    {
        const int col   = (wave * T::B_DMA + i) * B_COLS_PER_INSTR + lane / B_CHUNKS;
        const int chunk = (lane % B_CHUNKS) ^ ((col >> 2) & (B_CHUNKS - 1));
        b_src[i]        = (const uint8_t*)p.b + (size_t)(n0 + col) * p.ldb_bytes +
                   (size_t)kz_begin / 8 * BInfo<BKIND>::kBytesPer8K + chunk * 16;
    }
    // Pre-shuffled packed int4 (p.b_preshuf): the 32 columns x 64 k a wave moves per K tile are one contiguous KiB of
    // global memory, already in the order of the LDS image (unit (column 16 t + r, k half h) at 16-byte slot
    // 32 t + 16 h + r), so lane l copies slot l and consecutive K tiles are 1024 bytes apart.
    size_t b_tile_stride = T::B_ROW_BYTES;
    // per-lane byte offset, inside the B image of a K tile, of this lane's 4 bytes of k-step s of its column of
    // column block nt = 0 (further column blocks: + 512 nt in both layouts)
    int raw_off[2] = {0, 0};
    if constexpr(BKIND == B_I4P)
    {
#pragma unroll
        for(int s = 0; s < 2; ++s)
            raw_off[s] = (col_w + fr) * 32 + ((s ^ ((fr >> 2) & 1)) << 4) + fq * 4; // (col >> 2) & 1 == (fr >> 2) & 1
        if(p.b_preshuf)
        {
            b_src[0] = (const uint8_t*)p.b + (size_t)((n0 >> 5) + wave) * ((size_t)p.ldb_bytes * 32) +
                       (size_t)(kz_begin >> 6) * 1024 + lane * 16;
            b_tile_stride = 1024;
#pragma unroll
            for(int s = 0; s < 2; ++s)
                raw_off[s] = (col_w >> 5) * 1024 + (s * 16 + fr) * 16 + fq * 4;
        }
    }
    else
    {
        // Pre-shuffled 8-bit weights: a DMA piece (16 columns x 64 k) is one contiguous KiB whose 16-byte units sit
        // column after column, k quarter after k quarter; the lane that fills LDS slot l fetches the unit the swizzled
        // image wants there (column l / 4, k quarter (l % 4) ^ swizzle), so the image and its reads do not change.
        if(p.b_preshuf)
        {
#pragma unroll
            for(int i = 0; i < T::B_DMA; ++i)
            {
                const int c = lane >> 2, j = (lane & 3) ^ ((c >> 2) & 3);
                b_src[i]    = (const uint8_t*)p.b + (size_t)((n0 >> 4) + wave * T::B_DMA + i) * ((size_t)p.ldb_bytes * 16) +
                           (size_t)(kz_begin >> 6) * 1024 + (c * 4 + j) * 16;
            }
            b_tile_stride = 1024;
        }
    }

    auto stage = [&](int buf, int kt) __attribute__((always_inline)) {
        uint8_t* sa = lds + buf * T::STAGE;
        uint8_t* sb = sa + T::A_TILE;
#pragma unroll
        for(int i = 0; i < T::A_DMA; ++i)
            glds16(a_src[i] + (size_t)kt * T::A_ROW_BYTES, sa + (wave * T::A_DMA + i) * 1024);
#pragma unroll
        for(int i = 0; i < T::B_DMA; ++i)
            glds16(b_src[i] + (size_t)kt * b_tile_stride, sb + (wave * T::B_DMA + i) * 1024);
    };

    f4_t acc[NT][MT];
#pragma unroll
    for(int nt = 0; nt < NT; ++nt)
#pragma unroll
        for(int mt = 0; mt < MT; ++mt)
            acc[nt][mt] = f4_t{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS read offsets that do not depend on the buffer
    int a_off[2]; // k-step s: byte offset of this lane's 16-byte chunk in row (row_w + fr) of m-tile 0
    const int a_row0 = row_w + fr;
#pragma unroll
    for(int s = 0; s < 2; ++s)
        a_off[s] = a_row0 * 128 + (((4 * s + fq) ^ (a_row0 & 7)) << 4); // (row + 16 mt) & 7 == row & 7

    u4_t wf[NT], af[MT]; // operand fragments of one 32-deep k-step

    // L section: fetch this wave's fragments of k-step s from LDS and dequantise the weights.
    auto load_frags = [&](int buf, int s) __attribute__((always_inline)) {
        const uint8_t* sa = lds + buf * T::STAGE;
        const uint8_t* sb = sa + T::A_TILE;
        uint32_t raw4[NT];
        u2_t raw8[NT];
#pragma unroll
        for(int nt = 0; nt < NT; ++nt)
        {
            [[maybe_unused]] const int col = col_w + nt * 16 + fr;
            if constexpr(BKIND == B_I4P)
                raw4[nt] = *(const uint32_t*)(sb + raw_off[s] + nt * 512);
            else
                raw8[nt] = *(const u2_t*)(sb + col * 64 + (((2 * s + (fq >> 1)) ^ ((col >> 2) & 3)) << 4) +
                                          (fq & 1) * 8);
        }
#pragma unroll
        for(int mt = 0; mt < MT; ++mt)
            af[mt] = *(const u4_t*)(sa + a_off[s] + mt * 16 * 128);
#pragma unroll
        for(int nt = 0; nt < NT; ++nt)
        {
            if constexpr(BKIND == B_I4P)
            {
                if constexpr(AK == A_F16)
                    wf[nt] = dequant8_i4_f16(sc[nt], i4c, raw4[nt]);
                else
                    wf[nt] = dequant8(sc[nt], raw4[nt], std::integral_constant<int, B_I4P>{});
            }
            else
            {
                if constexpr(AK == A_F16)
                    wf[nt] = dequant8_bytes_f16<BKIND>(sc[nt], raw8[nt]);
                else
                    wf[nt] = dequant8_bytes_bf16<BKIND>(sc[nt], raw8[nt]);
            }
        }
    };

    // M section: MFMAs on the fragments in registers, accumulator tiles [mt_begin, mt_end).
    auto mfma_frags = [&](auto mt_begin, auto mt_end) __attribute__((always_inline)) {
#pragma unroll
        for(int mt = mt_begin; mt < mt_end; ++mt)
#pragma unroll
            for(int nt = 0; nt < NT; ++nt)
            {
                if constexpr(AK == A_F16)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        __builtin_bit_cast(h8_t, wf[nt]), __builtin_bit_cast(h8_t, af[mt]), acc[nt][mt], 0, 0, 0);
                else
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf8_t, wf[nt]), __builtin_bit_cast(bf8_t, af[mt]), acc[nt][mt], 0, 0, 0);
            }
    };
    constexpr std::integral_constant<int, 0> kMt0{};
    constexpr std::integral_constant<int, MT / 2> kMtHalf{};
    constexpr std::integral_constant<int, MT> kMtEnd{};

    // ---- main loop -----------------------------------------------------------------------------
    if(ntiles > 0)
    {
        stage(0, 0);
        if(ntiles > 1)
        {
            stage(1, 1);
            if constexpr(GROUPED)
                dma_scale(1);
            wait_vmcnt<T::DMA_PER_TILE + (GROUPED ? 1 : 0)>(); // tile 0 landed (this wave's part); tile 1 in flight
        }
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();

        if constexpr(SCHED == 5)
        {
            // Sections per K tile t and wave ("early" = first wave of a SIMD, "late" = second, one barrier behind):
            //   L0: DMA pieces [0, D0) of tile t+2; activation fragments of (t, k-step 0)                       b
            //   M0: raw weights of (t, 1); MFMAs of (t, 0) interleaved with their dequantisation;
            //       late waves: wait for tile t+1's DMA                                                           b
            //   L1: DMA pieces [D0, D) of tile t+2; activation fragments of (t, 1);
            //       early waves: wait for tile t+1's DMA                                                          b
            //   M1: raw weights of (t+1, 0); MFMAs of (t, 1) interleaved with their dequantisation               b
            // RAW: tile t+1 is first read in M1(t) (raw weights).  Early waves enter it behind barrier 4t+3 of
            //      their own count, late waves behind their 4t+4; every wave has waited for its own DMA of tile
            //      t+1 in the section that ends at ITS barrier 4t+3 (early: L1(t), late: M0(t)).
            // WAR: ring slot of tile t-1 is re-staged from L0(t) on; its last reads (L1(t-1), M0(t-1)) are
            //      retired by the lgkmcnt(0) in front of barriers that precede every L0(t).
            constexpr int D  = T::DMA_PER_TILE;
            constexpr int D0 = (D + 1) / 2;
            auto stage_part = [&](int buf, int kt, auto first, auto last) __attribute__((always_inline)) {
                uint8_t* sa = lds + buf * T::STAGE;
                uint8_t* sb = sa + T::A_TILE;
#pragma unroll
                for(int i = first; i < last; ++i)
                {
                    if(i < T::A_DMA)
                        glds16(a_src[i] + (size_t)kt * T::A_ROW_BYTES, sa + (wave * T::A_DMA + i) * 1024);
                    else
                        glds16(b_src[i - T::A_DMA] + (size_t)kt * b_tile_stride,
                               sb + (wave * T::B_DMA + (i - T::A_DMA)) * 1024);
                }
            };
            constexpr std::integral_constant<int, 0> kD_begin{};
            constexpr std::integral_constant<int, D0> kD_mid{};
            constexpr std::integral_constant<int, D> kD_end{};

            uint32_t raw4[NT];
            u2_t raw8[NT];
            auto read_raw = [&](int buf, int s) __attribute__((always_inline)) {
                const uint8_t* sb = lds + buf * T::STAGE + T::A_TILE;
#pragma unroll
                for(int nt = 0; nt < NT; ++nt)
                {
                    [[maybe_unused]] const int col = col_w + nt * 16 + fr;
                    if constexpr(BKIND == B_I4P)
                        raw4[nt] = *(const uint32_t*)(sb + raw_off[s] + nt * 512);
                    else
                        raw8[nt] = *(const u2_t*)(sb + col * 64 + (((2 * s + (fq >> 1)) ^ ((col >> 2) & 3)) << 4) +
                                                  (fq & 1) * 8);
                }
            };
            auto dequant_raw = [&](u4_t(&out)[NT]) __attribute__((always_inline)) {
#pragma unroll
                for(int nt = 0; nt < NT; ++nt)
                {
                    if constexpr(BKIND == B_I4P)
                    {
                        if constexpr(AK == A_F16)
                            out[nt] = dequant8_i4_f16(sc[nt], i4c, raw4[nt]);
                        else
                            out[nt] = dequant8(sc[nt], raw4[nt], std::integral_constant<int, B_I4P>{});
                    }
                    else
                    {
                        if constexpr(AK == A_F16)
                            out[nt] = dequant8_bytes_f16<BKIND>(sc[nt], raw8[nt]);
                        else
                            out[nt] = dequant8_bytes_bf16<BKIND>(sc[nt], raw8[nt]);
                    }
                }
            };
            auto load_a = [&](int buf, int s) __attribute__((always_inline)) {
                const uint8_t* sa = lds + buf * T::STAGE;
#pragma unroll
                for(int mt = 0; mt < MT; ++mt)
                    af[mt] = *(const u4_t*)(sa + a_off[s] + mt * 16 * 128);
            };
            auto mfma_with = [&](const u4_t(&w)[NT], auto mt_begin, auto mt_end) __attribute__((always_inline)) {
#pragma unroll
                for(int mt = mt_begin; mt < mt_end; ++mt)
#pragma unroll
                    for(int nt = 0; nt < NT; ++nt)
                    {
                        if constexpr(AK == A_F16)
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                __builtin_bit_cast(h8_t, w[nt]), __builtin_bit_cast(h8_t, af[mt]), acc[nt][mt], 0, 0, 0);
                        else
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf8_t, w[nt]), __builtin_bit_cast(bf8_t, af[mt]), acc[nt][mt], 0, 0, 0);
                    }
            };
            unsigned long long t_work[4] = {0, 0, 0, 0}, t_wait[4] = {0, 0, 0, 0}, t_a = 0;
            auto sync = [&](auto sec) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                unsigned long long t_b = 0;
                if constexpr(STAMP)
                {
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_b)::"memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();
                if constexpr(STAMP)
                {
                    unsigned long long t_c;
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_c)::"memory");
                    t_work[sec] += t_b - t_a;
                    t_wait[sec] += t_c - t_b;
                    t_a = t_c;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            constexpr std::integral_constant<int, 0> kL0{};
            constexpr std::integral_constant<int, 1> kM0{};
            constexpr std::integral_constant<int, 2> kL1{};
            constexpr std::integral_constant<int, 3> kM1{};

            // One MFMA section.  The first quarter of the MFMAs goes out at once; then the raw weights of the
            // next section are fetched from LDS, and the remaining MFMAs are issued in six groups with one
            // piece of that section's dequantisation behind each group (piece j = stage j / NT of fragment
            // j % NT for fp16 x int4; whole fragments for the other formats).
            constexpr int kSteps      = 6;
            constexpr int kMtPerStep  = (MT - MT / 4) / kSteps; // 2 (MT 16) or 1 (MT 8)
            constexpr int kPiecesStep = NT * 3 / kSteps;        // 1 (NT 2) or 2 (NT 4)
            static_assert(kMtPerStep * kSteps == MT - MT / 4 && kPiecesStep * kSteps == NT * 3);
            auto dequant_piece = [&](u4_t(&out)[NT], auto piece) __attribute__((always_inline)) {
                constexpr int j = decltype(piece)::value;
                constexpr int stage = j / NT, nt = j % NT;
                if constexpr(BKIND == B_I4P && AK == A_F16)
                    dequant8_i4_f16_stage<stage>(out[nt], sc[nt], i4c, raw4[nt]);
                else if constexpr(stage == 0)
                {
                    if constexpr(BKIND == B_I4P)
                        out[nt] = dequant8(sc[nt], raw4[nt], std::integral_constant<int, B_I4P>{});
                    else if constexpr(AK == A_F16)
                        out[nt] = dequant8_bytes_f16<BKIND>(sc[nt], raw8[nt]);
                    else
                        out[nt] = dequant8_bytes_bf16<BKIND>(sc[nt], raw8[nt]);
                }
                // pin the piece here: without an opaque use the optimiser sinks this arithmetic to the MFMAs
                // that consume it, i.e. into the head of the next MFMA section
                asm volatile("" : "+v"(out[nt][0]), "+v"(out[nt][1]), "+v"(out[nt][2]), "+v"(out[nt][3]));
            };
            auto mfma_step = [&](const u4_t(&w)[NT], u4_t(&w_next)[NT], auto dq_t, auto step) __attribute__((always_inline)) {
                constexpr int st  = decltype(step)::value;
                constexpr bool dq = decltype(dq_t)::value;
                mfma_with(w, std::integral_constant<int, MT / 4 + st * kMtPerStep>{},
                          std::integral_constant<int, MT / 4 + (st + 1) * kMtPerStep>{});
                if constexpr(dq)
                {
                    dequant_piece(w_next, std::integral_constant<int, st * kPiecesStep>{});
                    if constexpr(kPiecesStep == 2)
                        dequant_piece(w_next, std::integral_constant<int, st * kPiecesStep + 1>{});
                    // spread the VALU instructions of the piece(s) over the MFMA gaps of this group
#pragma unroll
                    for(int i = 0; i + 1 < kMtPerStep * NT; ++i)
                    {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mfma_section = [&](const u4_t(&w)[NT], u4_t(&w_next)[NT], auto dq, int raw_buf, int raw_s)
                __attribute__((always_inline)) {
                mfma_with(w, kMt0, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr(decltype(dq)::value)
                    read_raw(raw_buf, raw_s); // behind the first MFMA: its operand wait must not cover this read
                __builtin_amdgcn_sched_barrier(0);
                mfma_with(w, std::integral_constant<int, 1>{}, std::integral_constant<int, MT / 4>{});
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(w, w_next, dq, std::integral_constant<int, 0>{});
                mfma_step(w, w_next, dq, std::integral_constant<int, 1>{});
                mfma_step(w, w_next, dq, std::integral_constant<int, 2>{});
                mfma_step(w, w_next, dq, std::integral_constant<int, 3>{});
                mfma_step(w, w_next, dq, std::integral_constant<int, 4>{});
                mfma_step(w, w_next, dq, std::integral_constant<int, 5>{});
            };

            u4_t wf2[NT]; // weight fragments of the section after the current one
            read_raw(0, 0);
            dequant_raw(wf);
            __builtin_amdgcn_sched_barrier(0);
            if(late)
                __builtin_amdgcn_s_barrier();
            if constexpr(STAMP)
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_a)::"memory");
            int cur = 0;
            // one K tile; HAS_NEXT (compile time) = a tile t+1 exists, so M1 prepares its first weight fragments
            auto tile_body = [&](int t, auto has_next) __attribute__((always_inline)) {
                constexpr bool HAS_NEXT = decltype(has_next)::value;
                int nxt = cur + 1, nxt2 = cur + 2;
                nxt     = nxt >= T::NBUF ? nxt - T::NBUF : nxt;
                nxt2    = nxt2 >= T::NBUF ? nxt2 - T::NBUF : nxt2;
                const bool more2 = t + 2 < ntiles;
                // L0
                if(more2)
                {
                    stage_part(nxt2, t + 2, kD_begin, kD_mid);
                    if constexpr(GROUPED)
                        dma_scale(t + 2);
                }
                load_a(cur, 0);
                sync(kL0);
                // M0
                __builtin_amdgcn_s_setprio(1);
                mfma_section(wf, wf2, std::true_type{}, cur, 1);
                __builtin_amdgcn_s_setprio(0);
                if(late)
                {
                    if(more2)
                        wait_vmcnt<D0 + (GROUPED ? 1 : 0)>();
                    else
                        wait_vmcnt<0>();
                }
                sync(kM0);
                // L1
                if(more2)
                    stage_part(nxt2, t + 2, kD_mid, kD_end);
                load_a(cur, 1);
                if(!late)
                {
                    if(more2)
                        wait_vmcnt<D + (GROUPED ? 1 : 0)>();
                    else
                        wait_vmcnt<0>();
                }
                if constexpr(GROUPED && HAS_NEXT)
                {
                    // the scales of tile t+1 have landed (late waves waited in M0); fetched here, the read is
                    // covered by the lgkmcnt(0) of the section end
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for(int nt = 0; nt < NT; ++nt)
                        sc_next[nt] = *(const volatile uint16_t*)(sc_lds + ((t + 1) & 1) * 256 + (nt * 16 + fr) * 2);
                }
                sync(kL1);
                if constexpr(GROUPED && HAS_NEXT)
                {
#pragma unroll
                    for(int nt = 0; nt < NT; ++nt)
                        sc[nt].set(sc_next[nt]);
                }
                // M1
                __builtin_amdgcn_s_setprio(1);
                mfma_section(wf2, wf, std::integral_constant<bool, HAS_NEXT>{}, nxt, 0);
                __builtin_amdgcn_s_setprio(0);
                sync(kM1);
                cur = nxt;
            };
            for(int t = 0; t + 1 < ntiles; ++t)
                tile_body(t, std::true_type{});
            tile_body(ntiles - 1, std::false_type{});
            if(!late)
                __builtin_amdgcn_s_barrier();
            if constexpr(STAMP)
            {
                if(blockIdx.x == 0 && lane == 0 && p.dbg)
                {
                    unsigned long long* d = (unsigned long long*)p.dbg + wave * 8;
#pragma unroll
                    for(int i = 0; i < 4; ++i)
                    {
                        d[i]     = t_work[i];
                        d[4 + i] = t_wait[i];
                    }
                }
            }
        }
        else if constexpr(SCHED == 0)
        {
            int cur = 0; // buffer of tile t
            for(int t = 0; t < ntiles; ++t)
            {
                int nxt2 = cur + 2;
                nxt2     = nxt2 >= T::NBUF ? nxt2 - T::NBUF : nxt2;
                if(t + 2 < ntiles)
                    stage(nxt2, t + 2); // overwrites the buffer of tile t-1: every wave passed the last barrier after reading it
                load_frags(cur, 0);
                mfma_frags(kMt0, kMtEnd);
                load_frags(cur, 1);
                mfma_frags(kMt0, kMtEnd);
                // tile t+1 must have landed: leave only tile t+2's DMA (if it was issued) in flight
                if(t + 2 < ntiles)
                    wait_vmcnt<T::DMA_PER_TILE>();
                else
                    wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                cur = cur + 1 >= T::NBUF ? 0 : cur + 1;
            }
        }
        else
        {
            // Ping-pong: the second wave of every SIMD (waves 4..7) runs one barrier behind the first, so
            // between two consecutive barriers one wave of a SIMD issues its MFMAs while its partner reads
            // LDS and dequantises (VALU) for its next MFMA section.  Sections per K tile and wave:
            //   L0 [fragments of k-step 0]  b  M0  b  L1 [fragments of k-step 1; wait for tile t+1's DMA]
            //   b  M1  b,   with the DMA of tile t+2 issued in L0 (SCHED 1) or inside M1 (SCHED 2).
            // Every wave executes the same number of barriers (the late group one extra before the loop,
            // the early group one extra after it).
            //   RAW: a wave waits for its own DMA of tile t+1 at the end of L1(t); the first read of tile
            //        t+1 (early group's L0(t+1)) comes after at least one more barrier that every wave
            //        reaches only behind that wait.
            //   WAR: the buffer of tile t-1 is re-staged no earlier than L0(t) of the early group; the late
            //        group's last reads of it (its L1(t-1)) are retired by the lgkmcnt(0) in front of the
            //        barrier that the early group must pass before it starts L0(t).
            unsigned long long t_work[4] = {0, 0, 0, 0}, t_wait[4] = {0, 0, 0, 0}, t_a = 0;
            auto section_end = [&](auto sec) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                unsigned long long t_b = 0;
                if constexpr(STAMP)
                {
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_b)::"memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_barrier();
                if constexpr(STAMP)
                {
                    unsigned long long t_c;
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_c)::"memory");
                    t_work[sec] += t_b - t_a;
                    t_wait[sec] += t_c - t_b;
                    t_a = t_c;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            constexpr std::integral_constant<int, 0> kL0{};
            constexpr std::integral_constant<int, 1> kM0{};
            constexpr std::integral_constant<int, 2> kL1{};
            constexpr std::integral_constant<int, 3> kM1{};
            if(late)
                __builtin_amdgcn_s_barrier();
            if constexpr(STAMP)
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_a)::"memory");
            int cur = 0;
            for(int t = 0; t < ntiles; ++t)
            {
                int nxt2 = cur + 2;
                nxt2     = nxt2 >= T::NBUF ? nxt2 - T::NBUF : nxt2;
                // L0
                load_frags(cur, 0);
                section_end(kL0);
                // M0
                __builtin_amdgcn_s_setprio(1);
                mfma_frags(kMt0, kMtEnd);
                __builtin_amdgcn_s_setprio(0);
                section_end(kM0);
                // L1
                load_frags(cur, 1);
                wait_vmcnt<0>(); // only tile t+1's DMA (issued in M1(t-1) or the prologue) is outstanding
                section_end(kL1);
                // M1
                __builtin_amdgcn_s_setprio(1);
                {
                    mfma_frags(kMt0, kMtHalf);
                    __builtin_amdgcn_sched_barrier(0);
                    if(t + 2 < ntiles)
                        stage(nxt2, t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_frags(kMtHalf, kMtEnd);
                }
                __builtin_amdgcn_s_setprio(0);
                section_end(kM1);
                cur = cur + 1 >= T::NBUF ? 0 : cur + 1;
            }
            if(!late)
                __builtin_amdgcn_s_barrier();
            if constexpr(STAMP)
            {
                if(blockIdx.x == 0 && lane == 0 && p.dbg)
                {
                    unsigned long long* d = (unsigned long long*)p.dbg + wave * 8;
#pragma unroll
                    for(int i = 0; i < 4; ++i)
                    {
                        d[i]     = t_work[i];
                        d[4 + i] = t_wait[i];
                    }
                }
            }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // (every wave is past the last barrier of the loop: no LDS read or DMA of the ring is outstanding)
    tile256_epilogue<NT, MT>(p, acc, lds, m0, n0, row_w, col_w, fr, fq, wave, lane);
}

} // namespace wq
