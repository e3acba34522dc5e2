// Timing diagnostics for the fp8 x int4 256-tile kernel (not part of the product library): the kernel body with single
// ingredients of its K loop removed (template parameter PROBE of wq::f8s_kernel_body), each timed on the same operands.
// Results of the reduced variants are meaningless; the differences in time say what the loop waits for.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Icomposable_kernel_amd/csrc \
//         -o tools/f8_probe tools/f8_probe.hip && tools/f8_probe [K ...]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "wqgemm.h"
#include "wq_gemm_256_f8s.h"

using namespace wq;

template <int PROBE>
__global__ __launch_bounds__(512, 2) void probe_kernel(const GemmParams p)
{
    f8s_kernel_body<PROBE>(p);
}

#define CHECK(x)                                                                         \
    do                                                                                   \
    {                                                                                    \
        hipError_t e = (x);                                                              \
        if(e != hipSuccess)                                                              \
        {                                                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
            exit(1);                                                                     \
        }                                                                                \
    } while(0)

template <int PROBE>
float time_probe(const GemmParams& gp, dim3 grid, int iters)
{
    auto k = probe_kernel<PROBE>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Mfma256F8S::LDS_BYTES));
    for(int i = 0; i < 300; ++i)
        hipLaunchKernelGGL(k, grid, dim3(512), Mfma256F8S::LDS_BYTES, 0, gp);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    for(int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(k, grid, dim3(512), Mfma256F8S::LDS_BYTES, 0, gp);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv)
{
    std::vector<int> ks;
    for(int i = 1; i < argc; ++i)
        ks.push_back(atoi(argv[i]));
    if(ks.empty())
        ks = {128, 4096};
    const int M = 4096, N = 4096;
    for(int K : ks)
    {
        std::vector<uint8_t> a((size_t)M * K), b((size_t)N * K / 2);
        uint32_t s = 12345u;
        for(auto& x : a)
        {
            s = s * 1664525u + 1013904223u;
            x = (uint8_t)(24 + ((s >> 16) % 64) + ((s >> 30) & 1) * 128); // moderate E4M3 magnitudes, both signs
        }
        for(auto& x : b)
        {
            s = s * 1664525u + 1013904223u;
            x = (uint8_t)(s >> 20);
        }
        std::vector<float> d0(N), d1(M);
        for(auto& x : d0)
            x = 0.01f;
        for(auto& x : d1)
            x = 0.5f;
        void *da, *db, *dc, *dd0, *dd1;
        CHECK(hipMalloc(&da, a.size()));
        CHECK(hipMalloc(&db, b.size()));
        CHECK(hipMalloc(&dc, (size_t)M * N * 2));
        CHECK(hipMalloc(&dd0, N * 4));
        CHECK(hipMalloc(&dd1, M * 4));
        CHECK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dd0, d0.data(), N * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dd1, d1.data(), M * 4, hipMemcpyHostToDevice));
        GemmParams gp{};
        gp.a = da, gp.b = db, gp.scale = nullptr, gp.c = dc;
        gp.M = M, gp.N = N, gp.K = K, gp.lda = K, gp.ldb_bytes = K / 2, gp.ldc = N;
        gp.k_begin = 0, gp.k_end = K, gp.k_per_z = K, gp.partial = 0;
        gp.tiles_m = M / 256, gp.tiles_n = N / 256, gp.group_m = 4, gp.direct_store = 0;
        gp.a_scale = 0x7E7E7E7E;
        gp.epi.kind = 4, gp.epi.c_dtype = 0, gp.epi.d0 = dd0, gp.epi.d1 = dd1, gp.epi.ldd0 = 0, gp.epi.ldd1 = 0;
        gp.b_preshuf = 1; // (same bytes read as the pre-shuffled order: the address pattern is what matters here)
        const dim3 grid(gp.tiles_m * gp.tiles_n, 1, 1);
        const int iters = K >= 16384 ? 200 : 500;
        const int ntiles = K / 128;
        struct R { const char* name; float us; };
        std::vector<R> r;
        r.push_back({"full kernel", time_probe<0>(gp, grid, iters)});
        auto stamped = [&](auto probe, const char* name) {
            constexpr int P = decltype(probe)::value;
            long long* dbg;
            CHECK(hipMalloc(&dbg, 256 * 8 * 4 * 8));
            gp.dbg = dbg;
            const float us = time_probe<P>(gp, grid, iters);
            r.push_back({name, us});
            std::vector<long long> h(256 * 8 * 4);
            CHECK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            double s0 = 0, s1 = 0, s2 = 0;
            for(int i = 0; i < 256 * 8; ++i)
                s0 += h[i * 4], s1 += h[i * 4 + 1], s2 += h[i * 4 + 2];
            printf("K = %d %s: stamps (shader cycles, mean over 2048 waves): prologue %.0f, K loop %.0f (%.0f per tile), epilogue + store %.0f; "
                   "kernel %.2f us\n", K, name, s0 / 2048, s1 / 2048, s1 / 2048 / ntiles, s2 / 2048, us);
            gp.dbg = nullptr;
            CHECK(hipFree(dbg));
        };
        stamped(std::integral_constant<int, 4096>{}, "stamped build");
        printf("K = %d (%d K tiles of 128; ideal matrix-pipe time per tile at 2.4 GHz: 0.853 us)\n", K, ntiles);
        for(auto& x : r)
            printf("  %-58s %8.2f us   %7.1f TFLOP/s\n", x.name, x.us, 2.0 * M * N * K / x.us / 1e6);
        hipFree(da), hipFree(db), hipFree(dc), hipFree(dd0), hipFree(dd1);
    }
    // per-tile cost = (t(K2) - t(K1)) / (tiles2 - tiles1) follows from two K values
    return 0;
}
