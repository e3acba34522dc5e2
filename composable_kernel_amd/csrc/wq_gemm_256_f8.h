// 256 x 256 x 128 MFMA kernel for fp8 activations x packed int4 weights (gfx950), four-section ping-pong form
// (wq_gemm_f8_mfma_256_pp; kept for A/B runs and the section-cycle diagnostics -- the product default is the
// software-pipelined form in wq_gemm_256_f8s.h, which shares the helpers below).
//
// Arithmetic: C[m][n] = epilogue( sum_k float(a[m][k]) * float(q[k][n]) ), the per-channel scales of
// the weights arrive through the epilogue (MultiplyMultiply: d0[n], d1[m]), as in
// example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8.cpp:43-47,152-153.
//
// Both operands go to the matrix core as 8-bit floats (v_mfma_scale_f32_16x16x128_f8f6f4, 128 k per
// instruction, twice the fp16 rate):
//   * weights: q in -8..7 has at most three significant bits, so it is exactly representable in
//     E5M2, which is the upper byte of fp16(q).  The packed nibbles are turned into fp16 pairs with
//     the same magic-number arithmetic as the fp16 kernel and the upper bytes are gathered with one
//     v_perm_b32 per four weights: 11 VALU instructions per 8 weights.
//   * activations: bytes are handed to the matrix core as E4M3 (OCP "e4m3fn", the only E4M3 the
//     gfx950 matrix core decodes) without touching them.  For FNUZ bytes (bias 8 instead of 7:
//     include/ck/utility/f8_utils.hpp:186-267) every finite encoding except 0x7F / 0xFF decodes to
//     exactly twice its FNUZ value, which the E8M0 block scale of the instruction (2^-1) undoes for
//     free.  0x7F / 0xFF (FNUZ +-240, an OCP NaN) and 0x80 (FNUZ NaN, an OCP -0) cannot be expressed;
//     the host side only routes FNUZ data here when the caller vouches for their absence.
//
// Structure (tile ring, LDS-DMA, ping-pong wave schedule, LDS-staged epilogue) follows
// wq_gemm_mfma_256; a K tile is one MFMA deep, so the two MFMA sections of a tile split the 16
// accumulator row blocks instead of the k-steps:
//   L0: DMA pieces [0,3) of tile t+2; activation fragments of row blocks 0..7          b
//   M0: 16 MFMAs (row blocks 0..7, column block 0 first) with the last 3 of the 8 conversion pieces of
//       this tile's weights (they complete column block 1's fragment) between the first 8;
//       late waves: wait for tile t+1's DMA                                            b
//   L1: DMA pieces [3,6) of tile t+2; activation fragments of row blocks 8..15;
//       early waves: wait for tile t+1's DMA                                           b
//   M1: 16 MFMAs (row blocks 8..15); raw weights of tile t+1 read behind the first, conversion
//       pieces 0..4 of tile t+1 between the last 13                                    b
// (conversion piece = one 32-bit word of 8 nibbles -> 8 E5M2 bytes, 11 VALU instructions; 3-4 of them
// fit behind each 32-cycle MFMA)
#pragma once

#include "wq_gemm_256.h"

namespace wq {

typedef int i8v_t __attribute__((ext_vector_type(8)));

struct Mfma256F8
{
    static constexpr int BM = 256, BN = 256, BK = 128, THREADS = 512, NBUF = 3;
    static constexpr int A_ROW_BYTES = BK;      // 128
    static constexpr int B_ROW_BYTES = BK / 2;  // 64
    static constexpr int A_TILE      = BM * A_ROW_BYTES; // 32 KiB
    static constexpr int B_TILE      = BN * B_ROW_BYTES; // 16 KiB
    static constexpr int STAGE       = A_TILE + B_TILE;
    static constexpr int RING_BYTES  = NBUF * STAGE;     // 144 KiB
    static constexpr int LDS_BYTES   = RING_BYTES > kCTileBytes ? RING_BYTES : kCTileBytes;
    static constexpr int A_DMA       = A_TILE / 1024 / 8; // 4
    static constexpr int B_DMA       = B_TILE / 1024 / 8; // 2
    static constexpr int DMA_PER_TILE = A_DMA + B_DMA;
};

// 16-byte chunk swizzles (XOR on the chunk index inside a row), applied on the DMA source address.
// A rows are 128 B (8 chunks); a lane of the 16x16x128 MFMA reads chunks 2 fq and 2 fq + 1 of row fr.
// B columns are 64 B (4 chunks); a lane reads chunk fq of column fr.  Both make every 16-lane group
// of a ds_read_b128 hit 64 distinct banks.
__device__ __forceinline__ int f8_a_swz(int row) { return ((row >> 1) & 1) | (row & 4); }
__device__ __forceinline__ int f8_b_swz(int col) { return (col >> 2) & 2; }

// (w & mask) | magic with the mask in a scalar register
__device__ __forceinline__ uint32_t and_or_s(uint32_t w, uint32_t mask, uint32_t magic)
{
    // (mask and magic are opaque register values, see I4F8Consts::init: one v_and_or_b32)
    return (w & mask) | magic;
}

struct I4F8Consts
{
    uint32_t mask_lo, mask_hi, magic;
    h2_t k1032, k16th, km72;
    __device__ __forceinline__ void init()
    {
        mask_lo = 0x000F000Fu;
        mask_hi = 0x00F000F0u;
        magic   = 0x64006400u;
        asm volatile("" : "+s"(mask_lo), "+s"(mask_hi), "+v"(magic));
        k1032 = h2_t{(_Float16)1032.0f, (_Float16)1032.0f};
        k16th = h2_t{(_Float16)0.0625f, (_Float16)0.0625f};
        km72  = h2_t{(_Float16)-72.0f, (_Float16)-72.0f};
    }
};

// 8 packed int4 (one 32-bit word, k = 0..7 in the packed order) -> 8 E5M2 bytes in k order.
__device__ __forceinline__ void dequant8_i4_e5m2(int& o0, int& o1, const I4F8Consts& c, uint32_t w)
{
    const uint32_t w8 = w >> 8;
    const h2_t t0     = as_h2(and_or_s(w, c.mask_lo, c.magic)) - c.k1032;                                  // k 0, 1
    const h2_t t1     = __builtin_elementwise_fma(as_h2(and_or_s(w, c.mask_hi, c.magic)), c.k16th, c.km72);  // k 2, 3
    const h2_t t2     = as_h2(and_or_s(w8, c.mask_lo, c.magic)) - c.k1032;                                 // k 4, 5
    const h2_t t3     = __builtin_elementwise_fma(as_h2(and_or_s(w8, c.mask_hi, c.magic)), c.k16th, c.km72); // k 6, 7
    o0                = (int)__builtin_amdgcn_perm(as_u32(t1), as_u32(t0), 0x07050301u);
    o1                = (int)__builtin_amdgcn_perm(as_u32(t3), as_u32(t2), 0x07050301u);
}

// (Issuing the LDS-DMA pieces between the MFMAs that carry no conversion work instead of in the load sections
// was measured 6 % slower: a piece costs an MFMA section 100-130 cycles.)
template <bool STAMP = false>
__global__ __launch_bounds__(512, 2) void wq_gemm_f8_mfma_256_pp(const GemmParams p)
{
    using T = Mfma256F8;
    constexpr int MT = 16, NT = 2, MH = MT / 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int lane  = threadIdx.x & 63;
    const int wave  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr    = lane & 15;
    const int fq    = lane >> 4;
    const int late  = wave >> 2;
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
    const int scale_w = 0x7F7F7F7F;  // weights: 2^0
    const int scale_a = p.a_scale;   // activations: 2^0 (E4M3FN bytes) or 2^-1 (FNUZ bytes)

    // LDS-DMA sources: buffer resource (scalar) + running scalar byte offset per piece (advanced one K tile
    // per use) + one 32-bit per-lane offset per operand; no 64-bit vector address arithmetic.
    // A: wave-instruction i of wave w moves rows 8 (4 w + i) .. +7 (lane -> row lane / 8, chunk lane % 8); the
    // swizzle depends on bits 1 and 2 of the row only, so the lane offset is the same for every i.
    // B: instruction i moves columns 16 (2 w + i) .. +15 (lane -> column lane / 4, chunk lane % 4); bit 3 of the
    // column is bit 5 of the lane.
    // The activation resource ends with the last byte of row M - 1: the rows a partial last row tile has past M are
    // out of range, which a buffer load answers without touching memory (their results are never stored).
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)((uint32_t)(p.M - 1) * (uint32_t)p.lda + (uint32_t)p.K), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, 0x7FFFFFFF, 0x00020000);
    uint32_t a_run[T::A_DMA], b_run[T::B_DMA];
#pragma unroll
    for(int i = 0; i < T::A_DMA; ++i)
        a_run[i] = (uint32_t)(m0 + (wave * T::A_DMA + i) * 8) * (uint32_t)p.lda + kz_begin;
#pragma unroll
    for(int i = 0; i < T::B_DMA; ++i)
        b_run[i] = (uint32_t)(n0 + (wave * T::B_DMA + i) * 16) * (uint32_t)p.ldb_bytes + kz_begin / 2;
    const uint32_t a_lane = (uint32_t)(lane >> 3) * p.lda + (((lane & 7) ^ f8_a_swz(lane >> 3)) << 4);
    const uint32_t b_lane = (uint32_t)(lane >> 2) * p.ldb_bytes + (((lane & 3) ^ f8_b_swz(lane >> 2)) << 4);
    auto blds16 = [&](const __amdgpu_buffer_rsrc_t& rsrc, uint32_t voff, uint32_t soff, uint8_t* lds_wave_base)
        __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff,
                                                 soff, 0, 0);
    };

    // issue DMA pieces [first, last) of the next K tile not yet staged (tiles are staged in order, each piece
    // exactly once per tile), into ring slot buf
    auto stage_part = [&](int buf, auto first, auto last) __attribute__((always_inline)) {
        uint8_t* sa = lds + buf * T::STAGE;
        uint8_t* sb = sa + T::A_TILE;
#pragma unroll
        for(int i = first; i < last; ++i)
        {
            if(i < T::A_DMA)
            {
                blds16(a_rsrc, a_lane, a_run[i], sa + (wave * T::A_DMA + i) * 1024);
                a_run[i] += T::A_ROW_BYTES;
            }
            else
            {
                blds16(b_rsrc, b_lane, b_run[i - T::A_DMA], sb + (wave * T::B_DMA + (i - T::A_DMA)) * 1024);
                b_run[i - T::A_DMA] += T::B_ROW_BYTES;
            }
        }
    };
    constexpr int D  = T::DMA_PER_TILE;
    constexpr int D0 = D / 2;
    constexpr std::integral_constant<int, 0> kD_begin{};
    constexpr std::integral_constant<int, D0> kD_mid{};
    constexpr std::integral_constant<int, D> kD_end{};

    f4_t acc[NT][MT];
#pragma unroll
    for(int nt = 0; nt < NT; ++nt)
#pragma unroll
        for(int mt = 0; mt < MT; ++mt)
            acc[nt][mt] = f4_t{0.f, 0.f, 0.f, 0.f};

    int a_off[2]; // this lane's two 16-byte chunks of row fr of row block 0
#pragma unroll
    for(int h = 0; h < 2; ++h)
        a_off[h] = fr * 128 + (((2 * fq + h) ^ f8_a_swz(fr)) << 4);
    int b_off[NT];
#pragma unroll
    for(int nt = 0; nt < NT; ++nt)
    {
        const int col = col_w + nt * 16 + fr;
        b_off[nt]     = col * 64 + ((fq ^ f8_b_swz(col)) << 4);
    }

    u4_t af[MH][2]; // activation fragments of half the row blocks
    u4_t raw[NT];
    i8v_t wf[NT], wf2[NT];

    auto load_a = [&](int buf, int mt_first) __attribute__((always_inline)) {
        const uint8_t* sa = lds + buf * T::STAGE;
#pragma unroll
        for(int i = 0; i < MH; ++i)
#pragma unroll
            for(int h = 0; h < 2; ++h)
                af[i][h] = *(const u4_t*)(sa + a_off[h] + (mt_first + i) * 16 * 128);
    };
    auto read_raw = [&](int buf) __attribute__((always_inline)) {
        const uint8_t* sb = lds + buf * T::STAGE + T::A_TILE;
#pragma unroll
        for(int nt = 0; nt < NT; ++nt)
            raw[nt] = *(const u4_t*)(sb + b_off[nt]);
    };
    // Conversion piece j (0..7) = 32-bit word j % 4 of raw[j / 4] -> registers 2 (j % 4), +1 of out[j / 4]:
    // 11 VALU instructions.  The opaque use pins the piece where it is written: without it the optimiser
    // sinks the arithmetic to the MFMAs that consume it.
    auto convert_piece = [&](i8v_t(&out)[NT], auto piece) __attribute__((always_inline)) {
        constexpr int j = decltype(piece)::value, nt = j / 4, d = j % 4;
        int o0, o1;
        dequant8_i4_e5m2(o0, o1, i4c, raw[nt][d]);
        out[nt][2 * d]     = o0;
        out[nt][2 * d + 1] = o1;
        asm volatile("" : "+v"(out[nt][2 * d]), "+v"(out[nt][2 * d + 1]));
    };
    // pieces [0, 5) are converted in M1 (or the prologue), [5, 8) in the first half of the next M0
    auto mfma_one = [&](const i8v_t(&w)[NT], int i, int mt, int nt) __attribute__((always_inline)) {
        i8v_t a;
        a[0] = af[i][0][0], a[1] = af[i][0][1], a[2] = af[i][0][2], a[3] = af[i][0][3];
        a[4] = af[i][1][0], a[5] = af[i][1][1], a[6] = af[i][1][2], a[7] = af[i][1][3];
        acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w[nt], a, acc[nt][mt], 1 /* E5M2 */, 0 /* E4M3 */,
                                                                        0, scale_w, 0, scale_a);
    };
    auto mfma_range = [&](const i8v_t(&w)[NT], int mt_first, auto first, auto last) __attribute__((always_inline)) {
#pragma unroll
        for(int j = first; j < last; ++j)
            mfma_one(w, j >> 1, mt_first + (j >> 1), j & 1);
    };
    // M0 order: all row blocks with column block 0 first, then column block 1 (whose fragment is completed
    // during the first half); M1 order: row-block major.
    auto mfma_m0 = [&](const i8v_t(&w)[NT], auto first, auto last) __attribute__((always_inline)) {
#pragma unroll
        for(int j = first; j < last; ++j)
            mfma_one(w, j & 7, j & 7, j >> 3);
    };
    // k MFMAs [first, first + k) of `issue`, then one conversion piece, its VALU instructions spread behind the MFMAs
    auto step = [&](auto issue, i8v_t(&out)[NT], auto first, auto count, auto piece) __attribute__((always_inline)) {
        constexpr int f = decltype(first)::value, k = decltype(count)::value;
        issue(std::integral_constant<int, f>{}, std::integral_constant<int, f + k>{});
        convert_piece(out, piece);
#pragma unroll
        for(int i = 0; i < k; ++i)
        {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (11 + k - 1) / k, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // MFMAs have no side effects the optimiser knows of, so nothing keeps them from sinking below the
    // section barriers to the next use of their accumulators (which would keep both halves of the
    // activation fragments alive at once).  An opaque use of the accumulators ends the section.
    auto pin_acc = [&](int mt_first) __attribute__((always_inline)) {
#pragma unroll
        for(int i = 0; i < MH; ++i)
            asm volatile("" : "+v"(acc[0][mt_first + i]), "+v"(acc[1][mt_first + i]));
    };

    unsigned long long t_work[4] = {0, 0, 0, 0}, t_wait[4] = {0, 0, 0, 0}, t_sub[2] = {0, 0}, t_a = 0;
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
    constexpr std::integral_constant<int, 0> k0{};
    constexpr std::integral_constant<int, 1> k1{};
    constexpr std::integral_constant<int, 16> k16{};

    if(ntiles > 0)
    {
        stage_part(0, kD_begin, kD_end);
        if(ntiles > 1)
        {
            stage_part(1, kD_begin, kD_end);
            wait_vmcnt<D>();
        }
        else
            wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();

        read_raw(0);
        convert_piece(wf, std::integral_constant<int, 0>{});
        convert_piece(wf, std::integral_constant<int, 1>{});
        convert_piece(wf, std::integral_constant<int, 2>{});
        convert_piece(wf, std::integral_constant<int, 3>{});
        convert_piece(wf, std::integral_constant<int, 4>{});
        __builtin_amdgcn_sched_barrier(0);
        if(late)
            __builtin_amdgcn_s_barrier();
        if constexpr(STAMP)
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_a)::"memory");

        int cur = 0;
        // One K tile with the weight fragments w; HAS_NEXT: M1 also converts tile t+1's weights into w_next.
        // RAW / WAR reasoning as in wq_gemm_mfma_256 (SCHED 5): tile t+1 is first read in M1(t), behind the
        // barrier that follows the section in which every wave waited for its own DMA of tile t+1; the ring
        // slot of tile t-1 is re-staged from L0(t) on, after the lgkmcnt(0) + barrier that retired its reads.
        auto tile_body = [&](int t, auto has_next) __attribute__((always_inline)) {
            constexpr bool HAS_NEXT = decltype(has_next)::value;
            i8v_t(&w)[NT]      = wf;
            i8v_t(&w_next)[NT] = wf2;
            int nxt = cur + 1, nxt2 = cur + 2;
            nxt     = nxt >= T::NBUF ? nxt - T::NBUF : nxt;
            nxt2    = nxt2 >= T::NBUF ? nxt2 - T::NBUF : nxt2;
            const bool more2 = t + 2 < ntiles;
            // L0
            if(more2)
                stage_part(nxt2, kD_begin, kD_mid);
            if constexpr(STAMP)
            {
                unsigned long long t_x;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_x)::"memory");
                t_sub[0] += t_x - t_a;
                __builtin_amdgcn_sched_barrier(0);
            }
            load_a(cur, 0);
            if constexpr(STAMP)
            {
                unsigned long long t_x;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_memtime %0" : "=s"(t_x)::"memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                t_sub[1] += t_x - t_a;
                __builtin_amdgcn_sched_barrier(0);
            }
            sync(kL0);
            // M0
            __builtin_amdgcn_s_setprio(1);
            {
                // column block 0 for the 8 row blocks, with pieces 5..7 (the rest of column block 1's fragment)
                auto issue = [&](auto a, auto b) __attribute__((always_inline)) { mfma_m0(w, a, b); };
                using ic = std::integral_constant<int, 0>;
                step(issue, wf, ic{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 5>{});
                step(issue, wf, std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 6>{});
                step(issue, wf, std::integral_constant<int, 6>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 7>{});
                mfma_m0(w, std::integral_constant<int, 8>{}, k16);
            }
            pin_acc(0);
            __builtin_amdgcn_s_setprio(0);
            if(late)
            {
                // tile t+1 complete; leave what was issued for tile t+2 so far (D0 pieces) in flight
                if(more2)
                    wait_vmcnt<D0>();
                else
                    wait_vmcnt<0>();
            }
            sync(kM0);
            // L1
            if(more2)
                stage_part(nxt2, kD_mid, kD_end);
            load_a(cur, MH);
            if(!late)
            {
                if(more2)
                    wait_vmcnt<D>();
                else
                    wait_vmcnt<0>();
            }
            sync(kL1);
            // M1
            __builtin_amdgcn_s_setprio(1);
            if constexpr(HAS_NEXT)
            {
                mfma_range(w, MH, k0, k1);
                __builtin_amdgcn_sched_barrier(0);
                read_raw(nxt); // behind the first MFMA: its operand wait must not cover this read
                __builtin_amdgcn_sched_barrier(0);
                mfma_range(w, MH, k1, std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
                auto issue = [&](auto a, auto b) __attribute__((always_inline)) { mfma_range(w, MH, a, b); };
                step(issue, w_next, std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
                step(issue, w_next, std::integral_constant<int, 6>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
                step(issue, w_next, std::integral_constant<int, 9>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
                step(issue, w_next, std::integral_constant<int, 12>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
                step(issue, w_next, std::integral_constant<int, 14>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            }
            else
                mfma_range(w, MH, k0, k16);
            pin_acc(MH);
            __builtin_amdgcn_s_setprio(0);
            if constexpr(HAS_NEXT)
            {
                wf[0]    = wf2[0];
                wf[1][0] = wf2[1][0];
                wf[1][1] = wf2[1][1];
            }
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
                d[64]     = t_sub[0]; // L0: DMA pieces issued
                d[64 + 1] = t_sub[1]; // L0: fragment reads issued (s_memtime itself returns through lgkmcnt)
            }
        }
    }

    tile256_epilogue<NT, MT>(p, acc, lds, m0, n0, 0, col_w, fr, fq, wave, lane);
}

} // namespace wq
