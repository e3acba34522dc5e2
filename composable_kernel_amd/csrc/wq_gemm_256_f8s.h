// 256 x 256 x 128 MFMA kernel for fp8 activations x packed int4 weights (gfx950), software-pipelined form.
//
// Same arithmetic, operand formats and epilogue as the ping-pong form wq_gemm_f8_mfma_256_pp (wq_gemm_256_f8.h): E4M3 activation bytes
// and weights converted in registers to E5M2 feed the matrix core's 16x16x128 f8f6f4 instruction -- here in its plain
// form v_mfma_f32_16x16x128_f8f6f4 (no block scales: measured 2 - 3 % faster than v_mfma_scale_*, whose scale-load half
// takes an issue slot of its own); FNUZ bytes, which the matrix core reads as twice their value, get their 2^-1 on the
// accumulators after the K loop (exact).  What differs is how a wave's loads meet its MFMAs:
//
//   * no load sections: every wave streams its own operands between its own MFMAs.  The activation fragments of
//     row block mt + LEAD are fetched from LDS while row block mt is multiplied (a rolling window of LEAD + 1
//     fragments, 8 VGPRs each), the weights of the NEXT K tile are converted a piece at a time behind the MFMAs
//     of the current one, and the global loads of later tiles are issued in the last row blocks of a tile;
//   * the packed weights never pass through LDS: with the 1 x 8 wave grid each weight is needed by exactly one
//     wave, so a lane loads the 16 bytes (32 nibbles = its 32 k of one column) of each of its two columns
//     straight into registers, two K tiles ahead;
//   * LDS holds only activations: a ring of four 32 KiB slots filled by LDS-DMA three K tiles ahead;
//   * ONE s_barrier per K tile (not four): barrier(t), in front of row block 12 of tile t, is the point where
//     tile t+1 is complete for everybody (each wave first waits for its own DMA of tile t+1 with a counted
//     vmcnt) and where the slot of tile t-1 may be refilled (a wave that reaches it has multiplied most row
//     blocks of tile t, so its reads of tile t-1 retired long ago).
// The two waves of a SIMD are symmetric and compete for the matrix pipe instruction by instruction; nothing forces
// them into phases, so a wave that stalls on a load leaves the pipe to its partner.
//
// VMEM bookkeeping (in-order completion, counted waits).  Tile t issues B(t+2) x 2 (registers) in its first row
// block and A(t+3) x 4 (LDS-DMA) behind barrier(t).  barrier(t) needs A(t+1), issued at the end of tile t-2;
// issued after it: B(t+1) x 2, A(t+2) x 4, B(t+2) x 2 -> vmcnt(8).  Converting B(t+1) (all through tile t) needs the
// loads issued a whole tile earlier; behind them when tile t starts: A(t+2) x 4 -> vmcnt(4).
#pragma once

#include "wq_gemm_256_f8.h"

namespace wq {

struct Mfma256F8S
{
    static constexpr int BM = 256, BN = 256, BK = 128, THREADS = 512, NBUF = 4;
    static constexpr int A_ROW_BYTES = BK;               // 128
    static constexpr int A_TILE      = BM * A_ROW_BYTES; // 32 KiB
    static constexpr int RING_BYTES  = NBUF * A_TILE;    // 128 KiB
    static constexpr int LDS_BYTES   = RING_BYTES > kCTileBytes ? RING_BYTES : kCTileBytes;
    static constexpr int A_DMA       = A_TILE / 1024 / 8; // LDS-DMA instructions per wave per tile: 4
    static constexpr int LEAD        = 3;                 // row blocks between a fragment's fetch and its MFMAs (window of 4 divides 16)
    // block-scale tables (ABSCALE kernel): [K tile][2] for the two row halves, then [wave][K tile], one spare tile each
    static constexpr int MAX_TILES_ABSCALE = 640;
    static constexpr int TAB_LD            = MAX_TILES_ABSCALE + 1;
    static constexpr int TAB_A             = kCTileBytes > RING_BYTES ? kCTileBytes : RING_BYTES;
    static constexpr int TAB_B             = TAB_A + 2 * TAB_LD * 4;
    static constexpr int LDS_BYTES_ABSCALE = TAB_B + 8 * TAB_LD * 4;
    static_assert(LDS_BYTES_ABSCALE <= 160 * 1024, "the scale tables must fit behind the C tile image");
};

// PROBE = 0 is the product kernel.  Other values exist only for the timing diagnostics in tools/f8_probe.hip: each bit
// removes one ingredient of the K loop (results are then meaningless; only the time is of interest).
//   1 no weight conversion   2 no fragment reads from LDS   4 no global loads   8 no barrier   16 no MFMA
//   32 zero E8M0 scale operands (instruction-selection check)
//   64 LDS-DMA pieces of a tile issued four row blocks apart instead of back to back behind the barrier
//   128, 256, 384: other instruction orders inside a row block
//   512 fragment reads issued but never consumed (the MFMAs keep the fragments of the prologue)
//   1024 no LDS-DMA in the loop (weight loads stay)   2048 no weight loads in the loop (LDS-DMA stays)
//   8192 epilogue without global stores   16384 epilogue without arithmetic (conversion and stores only)
//   4096 cycle stamps (s_memtime) at entry, first row block, end of the K loop and exit -> p.dbg[workgroup][wave][4]
// ABSCALE: block-wise scales of both operands (GemmParams::sa / sb; one K tile = one 128-k scale block).  Every row block's
// MFMAs then start from zero and their sums are folded into the accumulators with the block's a_scale * b_scale (one fused
// multiply-add per accumulator register and K tile, issued one row block after the MFMAs so that it never waits for them); the scales of the workgroup's two 128-row halves and of each wave's
// 32 columns sit in two small LDS tables behind the C tile image, written in the prologue.  With finer activation blocks
// (ABSCALE = 2: per-token scales and the like) the scales of the tile's 256 rows for one K tile travel with that tile instead:
// one more LDS-DMA piece (4 bytes per lane, rows 64 (w & 3) .. + 63 by wave w; waves w and w + 4 write the same bytes) into a
// ring of four 1 KiB slots, and every row block reads its lane's row scale from there.
template <int PROBE, int ABSCALE = 0>
__device__ __forceinline__ void f8s_kernel_body(const GemmParams& p)
{
    // ABSCALE: 0 none, 1 one activation scale per 128-row half of the tile (scale_block_m a multiple of 128), 2 one per row
    // (any other scale_block_m: row r takes table row (m0 + r) / scale_block_m)
    constexpr bool ROWSCALE = ABSCALE == 2;
    constexpr int AD        = Mfma256F8S::A_DMA + (ROWSCALE ? 1 : 0); // LDS-DMA instructions per wave and tile
    using T = Mfma256F8S;
    constexpr int MT = 16, NT = 2, LEAD = T::LEAD, WIN = LEAD + 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    [[maybe_unused]] long long stamp0 = 0, stamp1 = 0, stamp2 = 0;
    if constexpr(PROBE & 4096)
        stamp0 = __builtin_amdgcn_s_memtime();

    const int lane  = threadIdx.x & 63;
    const int wave  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr    = lane & 15;
    const int fq    = lane >> 4;
    const int col_w = wave * (NT * 16);

    int tm, tn;
    {
        const int nwg = p.tiles_m * p.tiles_n;
        tile_coords(xcd_remap(blockIdx.x, nwg), p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    }
    const int m0 = tm * T::BM, n0 = tn * T::BN;

    const int kz_begin = p.k_begin + blockIdx.z * p.k_per_z;
    const int kz_end   = min(p.k_end, kz_begin + p.k_per_z);
    const int ntiles   = (kz_end - kz_begin) / T::BK;

    I4F8Consts i4c;
    i4c.init();
    [[maybe_unused]] const int scale_w = 0x7F7F7F7F; // (diagnostic variant 32 only) weights: 2^0
    [[maybe_unused]] const int scale_a = p.a_scale;  // (diagnostic variant 32 only) activations: 2^0 (E4M3FN) or 2^-1 (FNUZ)

    // Activations: buffer resource ending with the last byte of row M - 1 (rows a partial last row tile has past M
    // are out of range: answered without touching memory, results never stored); a running scalar offset per
    // DMA piece and one per-lane offset (the source-side swizzle depends on bits 1, 2 of the row only).
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)((uint32_t)(p.M - 1) * (uint32_t)p.lda + (uint32_t)p.K), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, 0x7FFFFFFF, 0x00020000);
    // The tile body always issues the loads of tiles t+2 / t+3 (straight-line code, constant vmcnt arithmetic); past
    // the last tile they go through a resource of zero records: out of range, answered with zeros, no memory access.
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0, 0x00020000);
    uint32_t a_run[T::A_DMA];
#pragma unroll
    for(int i = 0; i < T::A_DMA; ++i)
        a_run[i] = (uint32_t)(m0 + (wave * T::A_DMA + i) * 8) * (uint32_t)p.lda + kz_begin;
    const uint32_t a_lane = (uint32_t)(lane >> 3) * p.lda + (((lane & 7) ^ f8_a_swz(lane >> 3)) << 4);
    // Weights: lane (fr, fq) owns bytes [16 fq, 16 fq + 16) of the 64-byte K-tile row of columns col_w + 16 nt + fr
    // Pre-shuffled weights (p.b_preshuf; blocks of 32 columns x 64 k = 1 KiB, unit (column 16 t + r, k half h) at slot
    // 32 t + 16 h + r, blocks of one column block consecutive along K): the lane's 32 k of K tile `tile` are k half
    // fq % 2 of block 2 tile + fq / 2, i.e. a wave-wide load reads four runs of 256 bytes.
    uint32_t b_voff[NT];
#pragma unroll
    for(int nt = 0; nt < NT; ++nt)
        b_voff[nt] = p.b_preshuf ? (uint32_t)((n0 + col_w) >> 5) * ((uint32_t)p.ldb_bytes * 32) + (uint32_t)(fq >> 1) * 1024 +
                                       (uint32_t)((nt * 32 + (fq & 1) * 16 + fr) * 16)
                                 : (uint32_t)(n0 + col_w + nt * 16 + fr) * (uint32_t)p.ldb_bytes + fq * 16;
    const uint32_t b_base      = p.b_preshuf ? (uint32_t)(kz_begin >> 6) * 1024 : kz_begin / 2;
    const uint32_t b_tile_step = p.b_preshuf ? 2048 : T::BK / 2;

    auto dma_a = [&](const __amdgpu_buffer_rsrc_t& rsrc, int slot, int i) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + slot * T::A_TILE + (wave * T::A_DMA + i) * 1024),
                                                 16, a_lane, a_run[i], 0, 0);
        a_run[i] += T::A_ROW_BYTES;
    };
    // weights of K tile `tile` (scalar byte offset = the loads' soffset)
    auto load_b = [&](const __amdgpu_buffer_rsrc_t& rsrc, u4_t(&dst)[NT], int tile) __attribute__((always_inline)) {
        const uint32_t soff = __builtin_amdgcn_readfirstlane(b_base + (uint32_t)tile * b_tile_step);
#pragma unroll
        for(int nt = 0; nt < NT; ++nt)
            dst[nt] = __builtin_bit_cast(u4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, b_voff[nt], soff, 0));
    };

    f4_t acc[NT][MT];
    auto clear_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for(int nt = 0; nt < NT; ++nt)
#pragma unroll
            for(int mt = 0; mt < MT; ++mt)
                acc[nt][mt] = f4_t{0.f, 0.f, 0.f, 0.f};
    };

    // this lane's two 16-byte chunks of row fr of row block 0 (swizzled position inside the 128-byte row)
    int a_off[2];
#pragma unroll
    for(int h = 0; h < 2; ++h)
        a_off[h] = fr * 128 + (((2 * fq + h) ^ f8_a_swz(fr)) << 4);

    i8v_t af[WIN];     // rolling window of activation fragments
    i8v_t afd[WIN];    // (diagnostic variant 512 only: where unconsumed reads land)
    u4_t raw[2][NT];   // packed weights of tiles t+1 / t+2 (by tile parity)
    i8v_t wf[2][NT];   // converted weights of tiles t / t+1 (by tile parity)

    auto read_a = [&](int slot, int mt, auto win) __attribute__((always_inline)) {
        constexpr int w   = decltype(win)::value;
        const uint8_t* sa = lds + slot * T::A_TILE + mt * 16 * 128;
        const u4_t lo     = *(const u4_t*)(sa + a_off[0]);
        const u4_t hi     = *(const u4_t*)(sa + a_off[1]);
        af[w][0] = lo[0], af[w][1] = lo[1], af[w][2] = lo[2], af[w][3] = lo[3];
        af[w][4] = hi[0], af[w][5] = hi[1], af[w][6] = hi[2], af[w][7] = hi[3];
    };
    auto read_a_unused = [&](int slot, int mt, auto win) __attribute__((always_inline)) {
        constexpr int w   = decltype(win)::value;
        const uint8_t* sa = lds + slot * T::A_TILE + mt * 16 * 128;
        u4_t lo, hi;
        asm volatile("ds_read_b128 %0, %1" : "=v"(lo) : "v"((uint32_t)(uintptr_t)(sa + a_off[0])));
        asm volatile("ds_read_b128 %0, %1" : "=v"(hi) : "v"((uint32_t)(uintptr_t)(sa + a_off[1])));
        afd[w][0] = lo[0], afd[w][1] = lo[1], afd[w][2] = lo[2], afd[w][3] = lo[3];
        afd[w][4] = hi[0], afd[w][5] = hi[1], afd[w][6] = hi[2], afd[w][7] = hi[3];
    };
    // conversion piece j (0..7) = 32-bit word j % 4 of raw[.][j / 4] -> registers 2 (j % 4), +1 of out[j / 4]
    auto convert_piece = [&](i8v_t(&out)[NT], const u4_t(&in)[NT], auto piece) __attribute__((always_inline)) {
        constexpr int j = decltype(piece)::value, nt = j / 4, d = j % 4;
        int o0, o1;
        dequant8_i4_e5m2(o0, o1, i4c, in[nt][d]);
        out[nt][2 * d]     = o0;
        out[nt][2 * d + 1] = o1;
        asm volatile("" : "+v"(out[nt][2 * d]), "+v"(out[nt][2 * d + 1]));
    };

    // half a conversion piece: half h (0 / 1) of 32-bit word d of raw[.][nt] -> register 2 d + h of out[nt] (4 weights,
    // 5 or 6 VALU instructions)
    auto convert_half = [&](i8v_t(&out)[NT], const u4_t(&in)[NT], auto idx) __attribute__((always_inline)) {
        constexpr int j = decltype(idx)::value, nt = j / 8, d = (j % 8) / 2, h = j % 2;
        const uint32_t w = h ? in[nt][d] >> 8 : in[nt][d];
        const h2_t t0    = as_h2(and_or_s(w, i4c.mask_lo, i4c.magic)) - i4c.k1032;
        const h2_t t1    = __builtin_elementwise_fma(as_h2(and_or_s(w, i4c.mask_hi, i4c.magic)), i4c.k16th, i4c.km72);
        out[nt][2 * d + h] = (int)__builtin_amdgcn_perm(as_u32(t1), as_u32(t0), 0x07050301u);
        asm volatile("" : "+v"(out[nt][2 * d + h]));
    };

    if(ntiles <= 0)
    {
        clear_acc();
        tile256_epilogue<NT, MT>(p, acc, lds, m0, n0, 0, col_w, fr, fq, wave, lane);
        return;
    }

    // Row-scale form: the activation scales of K tile `tile` for rows 64 (wave & 3) + lane, one float per lane, by LDS-DMA
    // into slot `slot` of a four-slot ring at TAB_A (rows past M re-read row M - 1: never stored)
    [[maybe_unused]] __amdgpu_buffer_rsrc_t s_rsrc = null_rsrc;
    [[maybe_unused]] uint32_t s_lane = 0, s_run = 0;
    if constexpr(ROWSCALE)
    {
        s_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.sa, 0, 0x7FFFFFFF, 0x00020000);
        s_lane = (uint32_t)(min(m0 + 64 * (wave & 3) + lane, p.M - 1) / p.bm) * (uint32_t)p.ld_sa * 4u;
        s_run  = (uint32_t)(kz_begin / T::BK) * 4u;
    }
    auto dma_s = [&](const __amdgpu_buffer_rsrc_t& rsrc, int slot) __attribute__((always_inline)) {
        if constexpr(ROWSCALE)
        {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + T::TAB_A + slot * 1024 + (wave & 3) * 256),
                                                     4, s_lane, s_run, 0, 0);
            s_run += 4;
        }
    };

    [[maybe_unused]] const float* tab_a = (const float*)(lds + T::TAB_A);
    [[maybe_unused]] const float* tab_b = (const float*)(lds + T::TAB_B) + wave * T::TAB_LD;
    [[maybe_unused]] float s_lo = 0.f, s_hi = 0.f, s_lo_next = 0.f, s_hi_next = 0.f; // a_scale * b_scale of rows 0..127 / 128..255
    [[maybe_unused]] f4_t pend0[2], pend1[2]; // block sums of this and the previous row block (by row block parity), not yet folded in
    if constexpr(ABSCALE)
    {
        // FNUZ bytes are multiplied as twice their value: their 2^-1 rides on the activation scales
        const float half = p.a_scale != 0x7F7F7F7F ? 0.5f : 1.f;
        const int kt0    = kz_begin / T::BK;
        float* wa        = (float*)(lds + T::TAB_A);
        float* wb        = (float*)(lds + T::TAB_B);
        if constexpr(!ROWSCALE)
            for(int i = threadIdx.x; i < 2 * (ntiles + 1); i += T::THREADS)
            {
                const int tt = min(i >> 1, ntiles - 1), h = i & 1;
                const int mb = min(m0 + h * 128, p.M - 1) / p.bm;
                wa[i]        = p.sa[(size_t)mb * p.ld_sa + kt0 + tt] * half;
            }
        for(int i = threadIdx.x; i < 8 * (ntiles + 1); i += T::THREADS)
        {
            const int wv = i / (ntiles + 1), tt = min(i % (ntiles + 1), ntiles - 1);
            // (row-scale form: the activations' 2^-1 rides on the weight scales)
            wb[wv * T::TAB_LD + i % (ntiles + 1)] =
                p.sb[(size_t)(kt0 + tt) + (size_t)((n0 + wv * 32) / p.bn) * p.ld_sb] * (ROWSCALE ? half : 1.f);
        }
        // (visible to every wave behind the prologue's barrier below)
    }

    // ---- prologue: A(0), B(0), A(1), B(1), A(2); tile 0 complete and converted, first LEAD fragments fetched ----
    constexpr bool SPREAD = (PROBE & 64) != 0;
#pragma unroll
    for(int i = 0; i < T::A_DMA; ++i)
        dma_a(a_rsrc, 0, i);
    dma_s(s_rsrc, 0);
    load_b(b_rsrc, raw[0], 0);
    {
        const __amdgpu_buffer_rsrc_t a1 = ntiles > 1 ? a_rsrc : null_rsrc, b1 = ntiles > 1 ? b_rsrc : null_rsrc;
        const __amdgpu_buffer_rsrc_t a2 = ntiles > 2 ? a_rsrc : null_rsrc;
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t s1 = ntiles > 1 ? s_rsrc : null_rsrc, s2 = ntiles > 2 ? s_rsrc : null_rsrc;
        if constexpr(SPREAD)
        {
            // (the weights first: tile 0 waits for them with the loop's own count, which needs four operations behind them)
            load_b(b1, raw[1], 1);
#pragma unroll
            for(int i = 0; i < T::A_DMA; ++i)
                dma_a(a1, 1, i);
            dma_a(a2, 2, 0); // pieces 1 .. 3 of A(2) follow inside tile 0
        }
        else
        {
#pragma unroll
            for(int i = 0; i < T::A_DMA; ++i)
                dma_a(a1, 1, i);
            dma_s(s1, 1);
            if constexpr(ROWSCALE)
                __builtin_amdgcn_sched_barrier(0); // (issue order is part of the vmcnt arithmetic)
            load_b(b1, raw[1], 1);
            if constexpr(ROWSCALE)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for(int i = 0; i < T::A_DMA; ++i)
                dma_a(a2, 2, i);
            dma_s(s2, 2);
        }
    }
    // (everything the first loads do not need happens while they are in flight: the 128 accumulator registers are cleared here)
    __builtin_amdgcn_sched_barrier(0);
    clear_acc();
    wait_vmcnt<SPREAD ? T::A_DMA + 3 : 2 * AD + 2>(); // A(0), B(0) done
    __builtin_amdgcn_s_barrier();
    {
        convert_piece(wf[0], raw[0], std::integral_constant<int, 0>{});
        convert_piece(wf[0], raw[0], std::integral_constant<int, 1>{});
        convert_piece(wf[0], raw[0], std::integral_constant<int, 2>{});
        convert_piece(wf[0], raw[0], std::integral_constant<int, 3>{});
        convert_piece(wf[0], raw[0], std::integral_constant<int, 4>{});
        convert_piece(wf[0], raw[0], std::integral_constant<int, 5>{});
        convert_piece(wf[0], raw[0], std::integral_constant<int, 6>{});
        convert_piece(wf[0], raw[0], std::integral_constant<int, 7>{});
    }
    if constexpr(ROWSCALE)
        s_lo_next = tab_b[0]; // (s_lo = the wave's weight scale of the tile; the row scale comes from the ring)
    else if constexpr(ABSCALE)
    {
        const float sb0 = tab_b[0];
        s_lo_next = tab_a[0] * sb0, s_hi_next = tab_a[1] * sb0;
    }
    read_a(0, 0, std::integral_constant<int, 0>{});
    read_a(0, 1, std::integral_constant<int, 1>{});
    read_a(0, 2, std::integral_constant<int, 2>{});
    static_assert(LEAD == 3, "prologue fetches LEAD fragments");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr(PROBE & 4096)
        stamp1 = __builtin_amdgcn_s_memtime();

    // ---- one K tile; PAR = t & 1 selects the register sets at compile time.  Straight-line: loads of tiles past the
    // last one are issued through the null resource, the reads and the conversion for a tile that does not exist
    // work on stale bytes whose results are never used. ----
    int slot = 0; // ring slot of tile t
    int t    = 0;
    auto tile_body = [&](auto par) __attribute__((always_inline)) {
        constexpr int P = decltype(par)::value, Q = 1 - P;
        const int slot1 = (slot + 1) & 3, slot3 = (slot + 3) & 3;
        const int slot2 = (slot + 2) & 3;
        const __amdgpu_buffer_rsrc_t b2 = t + 2 < ntiles ? b_rsrc : null_rsrc;
        const __amdgpu_buffer_rsrc_t a2 = t + 2 < ntiles ? a_rsrc : null_rsrc;
        const __amdgpu_buffer_rsrc_t a3 = t + 3 < ntiles ? a_rsrc : null_rsrc;
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t s3 = t + 3 < ntiles ? s_rsrc : null_rsrc;
        [[maybe_unused]] const float* row_s = (const float*)(lds + T::TAB_A + slot * 1024) + fr; // row 16 mt + fr at [16 mt]

        auto row_block = [&](auto mt_c) __attribute__((always_inline)) {
            constexpr int mt = decltype(mt_c)::value;
            constexpr int w  = mt % WIN;          // window entry multiplied now
            constexpr int wn = (mt + LEAD) % WIN; // window entry fetched now
            if constexpr(mt == MT - LEAD)
            {
                // barrier(t): tile t+1 complete for everybody, slot of tile t-1 free
                // (behind the last piece of A(t+1) -- SPREAD: issued in row block 9 of tile t-1 -- came A(t+2) piece 0, B(t+2) x 2
                // and A(t+2) pieces 1 .. 3)
                if constexpr(!(PROBE & 4))
                    wait_vmcnt<SPREAD ? 6 : AD + 4>();
                if constexpr(!(PROBE & 8))
                    __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            // fetch the fragment LEAD row blocks ahead (of this tile, or the first ones of the next)
            if constexpr(PROBE & 512)
            {
                if constexpr(mt + LEAD < MT)
                    read_a_unused(slot, mt + LEAD, std::integral_constant<int, wn>{});
                else
                    read_a_unused(slot1, mt + LEAD - MT, std::integral_constant<int, wn>{});
            }
            else if constexpr(!(PROBE & 2))
            {
                if constexpr(mt + LEAD < MT)
                    read_a(slot, mt + LEAD, std::integral_constant<int, wn>{});
                else
                    read_a(slot1, mt + LEAD - MT, std::integral_constant<int, wn>{});
            }
            // global loads of later tiles: weights at once (no LDS hazard), activations behind barrier(t)
            if constexpr(!(PROBE & 4))
            {
                if constexpr(mt == 0)
                    wait_vmcnt<AD>(); // B(t+1) has landed (issued a tile ago); behind it: A(t+2) x 4 (and its row scales)
                if constexpr(mt == 0 && !(PROBE & 2048))
                    load_b(b2, raw[P], t + 2); // B(t+2): raw[P] held tile t's weights, converted during tile t-1
                if constexpr(PROBE & 1024)
                {
                }
                else if constexpr(SPREAD)
                {
                    // piece 0 of A(t+3) behind barrier(t) (its slot held tile t-1), pieces 1 .. 3 of A(t+2) -- whose slot
                    // barrier(t-1) freed -- four row blocks apart: eight waves' pieces at a time instead of thirty-two
                    if constexpr(mt == MT - LEAD)
                        dma_a(a3, slot3, 0);
                    if constexpr(mt == 1 || mt == 5 || mt == 9)
                        dma_a(a2, slot2, 1 + mt / 4);
                }
                else if constexpr(mt >= MT - LEAD)
                {
                    dma_a(a3, slot3, mt - (MT - LEAD));
                    if constexpr(mt == MT - 1)
                    {
                        dma_a(a3, slot3, T::A_DMA - 1);
                        dma_s(s3, slot3);
                    }
                }
            }
            if constexpr(ABSCALE)
            {
                // Every row block's MFMAs start from zero and write one of two register sets (by row block parity); the sums of
                // row block mt - 1 are folded into the accumulators behind this row block's MFMAs, a whole row block after they
                // were issued, so that the fused multiply-adds neither wait for a result nor hold back the next MFMAs (the last
                // row block of a tile folds its own: nothing crosses the tile boundary).
                constexpr int cur = mt & 1, prv = cur ^ 1;
                if constexpr(mt == 0)
                    s_lo = s_lo_next, s_hi = s_hi_next;
                // (row form: s_hi = the row scale read a row block ago)
                [[maybe_unused]] const float ps = mt == 0 ? 0.f : ROWSCALE ? s_hi * s_lo : (mt - 1) < MT / 2 ? s_lo : s_hi;
                if constexpr(ROWSCALE)
                    s_hi = row_s[16 * mt]; // this row block's row scale: used when its sums are folded in
                if constexpr(mt == MT / 2)
                {
                    // the scales of tile t + 1 (the tables hold one spare entry past the last tile)
                    const float sbn = tab_b[t + 1];
                    if constexpr(ROWSCALE)
                        s_lo_next = sbn;
                    else
                        s_lo_next = tab_a[2 * (t + 1)] * sbn, s_hi_next = tab_a[2 * (t + 1) + 1] * sbn;
                }
                const f4_t zero = {0.f, 0.f, 0.f, 0.f};
                pend0[cur] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[P][0], af[w], zero, 1, 0, 0, 0, 0, 0);
                pend1[cur] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[P][1], af[w], zero, 1, 0, 0, 0, 0, 0);
                if constexpr(mt > 0)
                {
                    const f4_t sv   = {ps, ps, ps, ps};
                    acc[0][mt - 1] = __builtin_elementwise_fma(pend0[prv], sv, acc[0][mt - 1]);
                    acc[1][mt - 1] = __builtin_elementwise_fma(pend1[prv], sv, acc[1][mt - 1]);
                }
                if constexpr(!(PROBE & 1))
                    convert_half(wf[Q], raw[Q], std::integral_constant<int, mt>{});
                if constexpr(mt == MT - 1) // (nothing is carried across the tile boundary)
                {
                    const float ls = ROWSCALE ? s_hi * s_lo : s_hi;
                    const f4_t sv  = {ls, ls, ls, ls};
                    acc[0][mt]    = __builtin_elementwise_fma(pend0[cur], sv, acc[0][mt]);
                    acc[1][mt]    = __builtin_elementwise_fma(pend1[cur], sv, acc[1][mt]);
                }
                // (both MFMAs first, the fragment reads and all of the row block's vector work in their shadow: measured best)
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            else if constexpr(!(PROBE & 16))
            {
                // The plain (unscaled) form of the instruction: zero scale operands select v_mfma_f32_16x16x128_f8f6f4, which
                // has no scale-load half and issues faster than v_mfma_scale_*; FNUZ activations get their 2^-1 after the loop.
                // (PROBE & 32: the scaled form with both scales in registers, for comparison)
                if constexpr(PROBE & 32)
                {
                    acc[0][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[P][0], af[w], acc[0][mt], 1, 0, 0, scale_w, 0, scale_a);
                    acc[1][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[P][1], af[w], acc[1][mt], 1, 0, 0, scale_w, 0, scale_a);
                }
                else
                {
                    acc[0][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[P][0], af[w], acc[0][mt], 1 /* E5M2 */, 0 /* E4M3 */, 0, 0, 0, 0);
                    acc[1][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[P][1], af[w], acc[1][mt], 1, 0, 0, 0, 0, 0);
                }
            }
            else
            {
                // (keep the operands alive so that the loads, reads and conversions are not optimised away)
                asm volatile("" ::"v"(wf[P][0]), "v"(wf[P][1]), "v"(af[w]));
            }
            // weights of tile t+1: half a piece (4 weights) per row block, its VALU work behind the two MFMAs
            if constexpr(!(PROBE & 1) && !ABSCALE)
                convert_half(wf[Q], raw[Q], std::integral_constant<int, mt>{});
            constexpr int ORDER = (PROBE >> 7) & 3; // (diagnostic: other placements of the conversion work)
            if constexpr(ABSCALE)
            {
            }
            else if constexpr(ORDER == 0)
            {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); // the two fragment reads
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            else if constexpr(ORDER == 1)
            {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            else if constexpr(ORDER == 2)
            {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            else
            {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        row_block(std::integral_constant<int, 0>{});
        row_block(std::integral_constant<int, 1>{});
        row_block(std::integral_constant<int, 2>{});
        row_block(std::integral_constant<int, 3>{});
        row_block(std::integral_constant<int, 4>{});
        row_block(std::integral_constant<int, 5>{});
        row_block(std::integral_constant<int, 6>{});
        row_block(std::integral_constant<int, 7>{});
        row_block(std::integral_constant<int, 8>{});
        row_block(std::integral_constant<int, 9>{});
        row_block(std::integral_constant<int, 10>{});
        row_block(std::integral_constant<int, 11>{});
        row_block(std::integral_constant<int, 12>{});
        row_block(std::integral_constant<int, 13>{});
        row_block(std::integral_constant<int, 14>{});
        row_block(std::integral_constant<int, 15>{});
        slot = slot1;
        ++t;
    };
    static_assert(MT % WIN == 0, "the window entry of a row block must not depend on the tile");
    for(;;)
    {
        tile_body(std::integral_constant<int, 0>{});
        if(t == ntiles)
            break;
        tile_body(std::integral_constant<int, 1>{});
        if(t == ntiles)
            break;
    }
    wait_vmcnt<0>(); // (null-resource loads of tiles past the end still count)
    if constexpr(PROBE & 4096)
        stamp2 = __builtin_amdgcn_s_memtime();
    if constexpr(PROBE & 512)
        asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(afd[0]), "v"(afd[1]), "v"(afd[2]), "v"(afd[3]));

    // FNUZ bytes were multiplied as the E4M3FN values of the same bits, i.e. twice their value: halve the sums (exact: a
    // power of two; the products inside the matrix core scale the same way, so the bits equal those of a 2^-1 block scale)
    if constexpr(!(PROBE & 32) && !ABSCALE)
        if(p.a_scale != 0x7F7F7F7F)
        {
#pragma unroll
            for(int nt = 0; nt < NT; ++nt)
#pragma unroll
                for(int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = acc[nt][mt] * 0.5f;
        }
    __syncthreads(); // every wave is done with the ring before it becomes the C tile image
    if constexpr((PROBE & 8192) != 0) // (diagnostic: the epilogue without its global stores)
        tile256_epilogue_lds_k<4, false, NT, MT, 16, 16, true>(p, acc, lds, m0, n0, 0, col_w, fr, fq, wave, lane);
    else if constexpr((PROBE & 16384) != 0) // (diagnostic: conversion and stores only)
        tile256_epilogue_lds_k<0, false, NT, MT>(p, acc, lds, m0, n0, 0, col_w, fr, fq, wave, lane);
    else if constexpr(PROBE != 0) // (diagnostic builds: one epilogue kind keeps the code small)
        tile256_epilogue_lds_k<4, false, NT, MT>(p, acc, lds, m0, n0, 0, col_w, fr, fq, wave, lane);
    else
        tile256_epilogue<NT, MT>(p, acc, lds, m0, n0, 0, col_w, fr, fq, wave, lane);
    if constexpr(PROBE & 4096)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the stores of C have been acknowledged
        const long long stamp3 = __builtin_amdgcn_s_memtime();
        if(lane == 0 && p.dbg)
        {
            long long* out = (long long*)p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
            out[0] = stamp1 - stamp0, out[1] = stamp2 - stamp1, out[2] = stamp3 - stamp2, out[3] = stamp0;
        }
    }
}

__global__ __launch_bounds__(512, 2) void wq_gemm_f8_mfma_256(const GemmParams p) { f8s_kernel_body<0>(p); }
__global__ __launch_bounds__(512, 2) void wq_gemm_f8_mfma_256_abscale(const GemmParams p) { f8s_kernel_body<0, 1>(p); }
__global__ __launch_bounds__(512, 2) void wq_gemm_f8_mfma_256_abscale_rows(const GemmParams p) { f8s_kernel_body<0, 2>(p); }

} // namespace wq
