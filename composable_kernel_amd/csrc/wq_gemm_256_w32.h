// 256 x 256 x 64 MFMA kernel for 16-bit activations x packed int4 weights built on the 32 x 32 x 16
// MFMA shape (gfx950).
//
// Same tile ring, LDS-DMA staging, ping-pong wave schedule and LDS-staged epilogue as wq_gemm_mfma_256
// (wq_gemm_256.h); what changes is the shape of the matrix instruction.  A v_mfma_f32_32x32x16 runs 32
// cycles and holds the vector issue port for 8 of them, so the dequantisation (13 VALU instructions
// per 8 weights) hides behind the MFMAs with room to spare, where the 16-cycle 16x16x32 form leaves room
// for two instructions per MFMA.
//
// Wave w owns all 256 rows x columns 32 w .. 32 w + 31: 8 accumulators of 32 x 32.  The weight fragment
// is the first operand: lane l supplies column n = l & 31, and - because the k order inside a K tile is
// free as long as both operands agree - k-step s (0..3) of lane half h = l >> 5 is mapped to the 8
// physical k of 16-byte chunk 4 h + s.  A lane's weights of one K tile are then the 32 consecutive k
// 32 h .. 32 h + 31 of its column: ONE 16-byte LDS read, word s of it feeding k-step s.
// Accumulator register v of lane l holds C[m = 32 mb + (l & 31)][n = 8 (v >> 2) + 4 (l >> 5) + (v & 3)].
#pragma once

#include "wq_gemm_256.h"

namespace wq {

typedef float f16v_t __attribute__((ext_vector_type(16)));

struct Mfma256W32
{
    static constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512, NBUF = 3;
    static constexpr int A_ROW_BYTES = BK * 2;           // 128
    static constexpr int B_ROW_BYTES = BK / 2;           // 32
    static constexpr int A_TILE      = BM * A_ROW_BYTES; // 32 KiB
    static constexpr int B_TILE      = BN * B_ROW_BYTES; // 8 KiB
    static constexpr int STAGE       = A_TILE + B_TILE;
    static constexpr int RING_BYTES  = NBUF * STAGE; // 120 KiB
    static constexpr int LDS_BYTES   = RING_BYTES > kCTileBytes ? RING_BYTES : kCTileBytes;
    static constexpr int A_DMA       = 4, B_DMA = 1, DMA_PER_TILE = 5;
};

// chunk swizzles: a 16-lane group of a ds_read_b128 holds 8 even and 8 odd rows (columns) of one chunk index
__device__ __forceinline__ int w32_a_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int w32_b_swz(int col) { return (col >> 3) & 1; }

template <int AK, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void wq_gemm_mfma_256_w32(const GemmParams p)
{
    using T = Mfma256W32;
    constexpr int MB = 8; // 32-row blocks
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int lane  = threadIdx.x & 63;
    const int wave  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr    = lane & 31;
    const int lh    = lane >> 5;
    const int late  = wave >> 2;
    const int col_w = wave * 32;

    int tm, tn;
    {
        const int nwg = p.tiles_m * p.tiles_n;
        tile_coords(xcd_remap(blockIdx.x, nwg), p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    }
    const int m0 = tm * T::BM, n0 = tn * T::BN;

    const int kz_begin = p.k_begin + blockIdx.z * p.k_per_z;
    const int kz_end   = min(p.k_end, kz_begin + p.k_per_z);
    const int ntiles   = (kz_end - kz_begin) / T::BK;

    ScaleReg<AK> sc;
    sc.set(p.scale ? ((const uint16_t*)p.scale)[n0 + col_w + lr] : unit_scale_bits<AK>());

    // LDS-DMA sources: buffer resource + running scalar byte offset per piece + 32-bit lane offsets.
    // A piece i of wave w moves rows 8 (4 w + i) .. +7 (lane -> row lane / 8, chunk lane % 8); the swizzle uses
    // row bits 1..3, bit 3 being the parity of the piece.  The B piece of wave w moves columns 32 w .. +31.
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, 0x7FFFFFFF, 0x00020000);
    uint32_t a_run[T::A_DMA], b_run;
#pragma unroll
    for(int i = 0; i < T::A_DMA; ++i)
        a_run[i] = ((uint32_t)(m0 + (wave * T::A_DMA + i) * 8) * (uint32_t)p.lda + kz_begin) * 2;
    b_run = (uint32_t)(n0 + wave * 32) * (uint32_t)p.ldb_bytes + kz_begin / 2;
    uint32_t a_lane[2];
#pragma unroll
    for(int par = 0; par < 2; ++par)
        a_lane[par] = (uint32_t)(lane >> 3) * p.lda * 2 + (((lane & 7) ^ w32_a_swz(8 * par + (lane >> 3))) << 4);
    const uint32_t b_lane = (uint32_t)(lane >> 1) * p.ldb_bytes + (((lane & 1) ^ w32_b_swz(lane >> 1)) << 4);

    auto blds16 = [&](const __amdgpu_buffer_rsrc_t& rsrc, uint32_t voff, uint32_t soff, uint8_t* lds_wave_base)
        __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff,
                                                 soff, 0, 0);
    };
    // issue DMA pieces [first, last) of the next K tile not yet staged into ring slot buf
    auto stage_part = [&](int buf, auto first, auto last) __attribute__((always_inline)) {
        uint8_t* sa = lds + buf * T::STAGE;
        uint8_t* sb = sa + T::A_TILE;
#pragma unroll
        for(int i = first; i < last; ++i)
        {
            if(i < T::A_DMA)
            {
                blds16(a_rsrc, a_lane[i & 1], a_run[i], sa + (wave * T::A_DMA + i) * 1024);
                a_run[i] += T::A_ROW_BYTES;
            }
            else
            {
                blds16(b_rsrc, b_lane, b_run, sb + wave * 1024);
                b_run += T::B_ROW_BYTES;
            }
        }
    };
    constexpr int D = T::DMA_PER_TILE, D0 = 3;
    constexpr std::integral_constant<int, 0> kD_begin{};
    constexpr std::integral_constant<int, D0> kD_mid{};
    constexpr std::integral_constant<int, D> kD_end{};

    f16v_t acc[MB];
#pragma unroll
    for(int mb = 0; mb < MB; ++mb)
#pragma unroll
        for(int v = 0; v < 16; ++v)
            acc[mb][v] = 0.f;

    int a_off[4]; // k-step s: this lane's chunk of row lr of row block 0
#pragma unroll
    for(int s = 0; s < 4; ++s)
        a_off[s] = lr * 128 + (((4 * lh + s) ^ w32_a_swz(lr)) << 4);
    const int b_off = (col_w + lr) * 32 + ((lh ^ w32_b_swz(lr)) << 4); // (col_w is a multiple of 32)

    u4_t af[MB][2]; // activation fragments of two k-steps
    u4_t raw;       // this lane's 32 packed weights of one K tile
    u4_t wf[2], wf2[2];

    auto load_a = [&](int buf, int s_first) __attribute__((always_inline)) {
        const uint8_t* sa = lds + buf * T::STAGE;
#pragma unroll
        for(int mb = 0; mb < MB; ++mb)
#pragma unroll
            for(int j = 0; j < 2; ++j)
                af[mb][j] = *(const u4_t*)(sa + a_off[s_first + j] + mb * 32 * 128);
    };
    auto read_raw = [&](int buf) __attribute__((always_inline)) {
        raw = *(const u4_t*)(lds + buf * T::STAGE + T::A_TILE + b_off);
    };
    // dequantise word s of the raw weights: the 8 weights of k-step s
    auto dequant_word = [&](u4_t& out, int s) __attribute__((always_inline)) {
        out = dequant8(sc, raw[s], std::integral_constant<int, B_I4P>{});
        // pin: without an opaque use the optimiser sinks the arithmetic to the consuming MFMAs
        asm volatile("" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]));
    };
    auto mfma_one = [&](const u4_t& w, const u4_t& a, int mb) __attribute__((always_inline)) {
        if constexpr(AK == A_F16)
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, w), __builtin_bit_cast(h8_t, a),
                                                             acc[mb], 0, 0, 0);
        else
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8_t, w), __builtin_bit_cast(bf8_t, a),
                                                              acc[mb], 0, 0, 0);
    };
    // MFMAs [first, last) of a section in k-step-major order: j -> (k-step j / 8, row block j % 8)
    auto mfma_range = [&](const u4_t(&w)[2], auto first, auto last) __attribute__((always_inline)) {
#pragma unroll
        for(int j = first; j < last; ++j)
            mfma_one(w[j >> 3], af[j & 7][j >> 3], j & 7);
    };
    // k MFMAs, then the dequantisation of one word spread behind them
    auto step = [&](const u4_t(&w)[2], auto first, auto count, u4_t& out, int word) __attribute__((always_inline)) {
        constexpr int f = decltype(first)::value, k = decltype(count)::value;
        mfma_range(w, std::integral_constant<int, f>{}, std::integral_constant<int, f + k>{});
        dequant_word(out, word);
#pragma unroll
        for(int i = 0; i < k; ++i)
        {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for(int mb = 0; mb < MB; ++mb)
            asm volatile("" : "+v"(acc[mb]));
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
    constexpr std::integral_constant<int, 0> k0{};
    constexpr std::integral_constant<int, 1> k1{};
    constexpr std::integral_constant<int, 4> k4{};
    constexpr std::integral_constant<int, 6> k6{};
    constexpr std::integral_constant<int, 8> k8{};
    constexpr std::integral_constant<int, 10> k10{};
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
        dequant_word(wf[0], 0);
        dequant_word(wf[1], 1);
        __builtin_amdgcn_sched_barrier(0);
        if(late)
            __builtin_amdgcn_s_barrier();
        if constexpr(STAMP)
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_a)::"memory");

        int cur = 0;
        // Sections of K tile t (early waves = first of their SIMD; late waves run one barrier behind):
        //   L0: DMA pieces [0,3) of tile t+2; activation fragments of k-steps 0, 1                               b
        //   M0: 16 MFMAs of k-steps 0, 1, dequantisation of k-steps 2, 3 between them; late waves: wait for
        //       tile t+1's DMA                                                                                     b
        //   L1: DMA pieces [3,5) of tile t+2; activation fragments of k-steps 2, 3; early waves: wait for t+1     b
        //   M1: 16 MFMAs of k-steps 2, 3; raw weights of tile t+1 read behind the first, their k-steps 0, 1
        //       dequantised between the rest                                                                      b
        // RAW / WAR reasoning as in wq_gemm_mfma_256 (SCHED 5).
        auto tile_body = [&](int t, auto has_next) __attribute__((always_inline)) {
            constexpr bool HAS_NEXT = decltype(has_next)::value;
            int nxt = cur + 1, nxt2 = cur + 2;
            nxt     = nxt >= T::NBUF ? nxt - T::NBUF : nxt;
            nxt2    = nxt2 >= T::NBUF ? nxt2 - T::NBUF : nxt2;
            const bool more2 = t + 2 < ntiles;
            // L0
            if(more2)
                stage_part(nxt2, kD_begin, kD_mid);
            load_a(cur, 0);
            sync(kL0);
            // M0
            __builtin_amdgcn_s_setprio(1);
            step(wf, k0, k8, wf2[0], 2);
            step(wf, k8, k8, wf2[1], 3);
            pin_acc();
            __builtin_amdgcn_s_setprio(0);
            if(late)
            {
                if(more2)
                    wait_vmcnt<D0>();
                else
                    wait_vmcnt<0>();
            }
            sync(kM0);
            // L1
            if(more2)
                stage_part(nxt2, kD_mid, kD_end);
            load_a(cur, 2);
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
                mfma_range(wf2, k0, k1);
                __builtin_amdgcn_sched_barrier(0);
                read_raw(nxt); // behind the first MFMA: its operand wait must not cover this read
                __builtin_amdgcn_sched_barrier(0);
                mfma_range(wf2, k1, k4);
                __builtin_amdgcn_sched_barrier(0);
                step(wf2, k4, k6, wf[0], 0);
                step(wf2, k10, k6, wf[1], 1);
            }
            else
                mfma_range(wf2, k0, k16);
            pin_acc();
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

    // accumulator register group g = v >> 2 of row block mb: 4 consecutive n of row 32 mb + lr
    f4_t acc4[4][MB];
#pragma unroll
    for(int g = 0; g < 4; ++g)
#pragma unroll
        for(int mb = 0; mb < MB; ++mb)
            acc4[g][mb] = f4_t{acc[mb][4 * g], acc[mb][4 * g + 1], acc[mb][4 * g + 2], acc[mb][4 * g + 3]};
    tile256_epilogue<4, MB, 32, 8>(p, acc4, lds, m0, n0, 0, col_w, lr, lh, wave, lane);
}

} // namespace wq
