// How many vector instructions fit in the shadow of one v_mfma_scale_f32_16x16x128_f8f6f4 (32 cycles of matrix pipe)?
// Register-only loops: per MFMA, NV independent VALU instructions of one kind; one or two waves per SIMD.
// Prints cycles per MFMA (s_memtime around the loop, median over waves is not needed: every wave runs the same).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i8v_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

// KIND: 0 v_and_or_b32, 1 v_pk_add_f16, 2 v_pk_fma_f16, 3 v_perm_b32, 4 the kernel's mix (and_or, and_or, pk_add, pk_fma, perm, lshr),
//       5 s_nop 0, 6 ds_read_b128 (conflict-free, own 1 KiB per wave region)
template <int NV, int KIND, int SCALED>
__global__ __launch_bounds__(512, 2) void probe(long long* out, int iters, int seed)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    f4_t acc[8];
    i8v_t a, b;
    for(int i = 0; i < 8; ++i)
        acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
    for(int i = 0; i < 8; ++i)
        a[i] = 0x38383838 + seed * (threadIdx.x + i), b[i] = 0x3C3C3C3C + seed * (threadIdx.x * 3 + i);
    unsigned v[8];
    for(int i = 0; i < 8; ++i)
        v[i] = threadIdx.x * 77 + i + seed;
    unsigned m1 = 0x000F000Fu, magic = 0x64006400u, sel = 0x07050301u;
    asm volatile("" : "+s"(m1), "+v"(magic), "+s"(sel));
    const int sa = 0x7F7F7F7F + seed, sb = 0x7F7F7F7F + seed;
    const unsigned lds_off = (threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 16;
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    u4_t ld[4];
    for(int i = 0; i < 4; ++i)
        ld[i] = u4_t{0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for(int it = 0; it < iters; ++it)
    {
#pragma unroll
        for(int u = 0; u < 8; ++u)
        {
            // (inline assembly pins the position of the MFMA among the fillers; independent accumulators, so no hazard)
            if constexpr(SCALED)
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:1"
                             : "+v"(acc[u]) : "v"(a), "v"(b), "v"(sa), "v"(sb));
            else
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:1" : "+v"(acc[u]) : "v"(a), "v"(b));
#pragma unroll
            for(int j = 0; j < NV; ++j)
            {
                unsigned& x = v[(u * NV + j) % 8];
                const int kind = KIND == 4 ? (j % 6) : KIND;
                if(kind == 0 || (KIND == 4 && kind == 1))
                    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "s"(m1), "v"(magic));
                else if(KIND == 1 || (KIND == 4 && kind == 2))
                    asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(magic));
                else if(KIND == 2 || (KIND == 4 && kind == 3))
                    asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(x) : "v"(magic));
                else if(KIND == 3 || (KIND == 4 && kind == 4))
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(magic), "s"(sel));
                else if(KIND == 4 && kind == 5)
                    asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(x));
                else if(KIND == 5)
                    asm volatile("s_nop 0");
                else if(KIND == 6)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j % 4]) : "v"(lds_off), "n"((j % 4) * 1024));
            }
        }
        if constexpr(KIND == 6)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for(int i = 0; i < 8; ++i)
        s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned vs = 0;
    for(int i = 0; i < 8; ++i)
        vs ^= v[i];
    for(int i = 0; i < 4; ++i)
        vs ^= ld[i][0] ^ ld[i][3];
    if(threadIdx.x % 64 == 0)
        out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
    if(s == 1.2345f && vs == 77)
        out[0] = (long long)lds[threadIdx.x];
}

template <int NV, int KIND, int SCALED>
void run(const char* what, int threads)
{
    long long* d;
    hipMalloc(&d, 256 * 8 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((probe<NV, KIND, SCALED>), dim3(256), dim3(threads), 0, 0, d, 200, 0);
    hipLaunchKernelGGL((probe<NV, KIND, SCALED>), dim3(256), dim3(threads), 0, 0, d, iters, 0);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    const int waves = threads / 64;
    for(int b = 0; b < 256; ++b)
        for(int w = 0; w < waves; ++w)
            sum += (double)h[b * 8 + w];
    const double per_wave = sum / (256 * waves) / (iters * 8.0);   // cycles per MFMA of one wave
    const double per_simd = per_wave / (waves / 4.0);              // cycles per MFMA issued on a SIMD
    printf("%-34s NV=%d %s waves/SIMD=%d: %6.1f cycles per MFMA per SIMD\n", what, NV, SCALED ? "scaled  " : "unscaled", waves / 4, per_simd);
    hipFree(d);
}

#define ROW(KIND, NAME)                                                    \
    run<0, KIND, 1>(NAME, T), run<1, KIND, 1>(NAME, T), run<2, KIND, 1>(NAME, T), run<3, KIND, 1>(NAME, T), \
        run<4, KIND, 1>(NAME, T), run<6, KIND, 1>(NAME, T), run<8, KIND, 1>(NAME, T)

int main()
{
    for(int T : {256, 512})
    {
        run<0, 0, 0>("bare", T);
        ROW(0, "v_and_or_b32");
        ROW(1, "v_pk_add_f16");
        ROW(2, "v_pk_fma_f16");
        ROW(3, "v_perm_b32");
        ROW(4, "kernel's conversion mix");
        ROW(5, "s_nop 0");
        run<1, 6, 1>("ds_read_b128", T), run<2, 6, 1>("ds_read_b128", T), run<4, 6, 1>("ds_read_b128", T);
    }
    return 0;
}
