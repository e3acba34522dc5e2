// v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 beside vector and LDS instructions: cycles per MFMA per SIMD
// (s_memtime), one and two waves per SIMD, register operands, independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

// SHAPE 0: 16x16x32, 1: 32x32x16.  NV VALU (v_pk_mul_f16) and ND ds_read_b128 per MFMA (ND2 = reads per two MFMAs).
template <int SHAPE, int NV, int ND2>
__global__ __launch_bounds__(512, 2) void probe(long long* out, int iters, int seed)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    h8_t a, b;
    for(int i = 0; i < 8; ++i)
        a[i] = (_Float16)(1.f + 0.001f * (threadIdx.x + i + seed)), b[i] = (_Float16)(0.5f + 0.002f * (threadIdx.x * 3 + i));
    f4_t acc4[8];
    f16_t acc16[4];
    for(int i = 0; i < 8; ++i)
        acc4[i] = f4_t{0, 0, 0, 0};
    for(int i = 0; i < 4; ++i)
        for(int j = 0; j < 16; ++j)
            acc16[i][j] = 0.f;
    unsigned v[8];
    for(int i = 0; i < 8; ++i)
        v[i] = 0x3C003C00u + threadIdx.x + i;
    unsigned c = 0x3C013C01u;
    asm volatile("" : "+v"(c));
    const unsigned lds_off = (threadIdx.x >> 6) * 4096 + (threadIdx.x & 63) * 16;
    u4_t ld[4] = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for(int it = 0; it < iters; ++it)
    {
#pragma unroll
        for(int u = 0; u < 8; ++u)
        {
            if constexpr(SHAPE == 0)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[u]) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[u & 3]) : "v"(a), "v"(b));
#pragma unroll
            for(int j = 0; j < NV; ++j)
                asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[(u * NV + j) & 7]) : "v"(c));
            if constexpr(ND2 > 0)
                if((u & 1) == 0)
                {
#pragma unroll
                    for(int j = 0; j < ND2; ++j)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j & 3]) : "v"(lds_off), "n"((j & 3) * 1024));
                }
        }
        if constexpr(ND2 > 0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for(int i = 0; i < 8; ++i)
        s += acc4[i][0] + acc4[i][3];
    for(int i = 0; i < 4; ++i)
        s += acc16[i][0] + acc16[i][15];
    unsigned vs = 0;
    for(int i = 0; i < 8; ++i)
        vs ^= v[i];
    for(int i = 0; i < 4; ++i)
        vs ^= ld[i][0] ^ ld[i][3];
    if(threadIdx.x % 64 == 0)
        out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
    if(s == 1.2345f && vs == 77)
        out[0] = lds[threadIdx.x];
}

template <int SHAPE, int NV, int ND2>
void run()
{
    long long* d;
    hipMalloc(&d, 256 * 8 * 8);
    for(int threads : {256, 512})
    {
        const int iters = 2000;
        hipLaunchKernelGGL((probe<SHAPE, NV, ND2>), dim3(256), dim3(threads), 0, 0, d, 100, 0);
        hipLaunchKernelGGL((probe<SHAPE, NV, ND2>), dim3(256), dim3(threads), 0, 0, d, iters, 0);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 8);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        const int waves = threads / 64;
        for(int b = 0; b < 256; ++b)
            for(int w = 0; w < waves; ++w)
                sum += (double)h[b * 8 + w];
        const double per_wave = sum / (256 * waves) / (iters * 8.0);
        printf("%s  VALU/MFMA=%d  ds_read_b128 per 2 MFMA=%d  waves/SIMD=%d: %6.1f cycles per MFMA per SIMD (%.1f per 32 flop-units)\n",
               SHAPE ? "32x32x16_f16" : "16x16x32_f16", NV, ND2, waves / 4, per_wave / (waves / 4.0),
               per_wave / (waves / 4.0) * (SHAPE ? 0.5 : 1.0));
    }
    hipFree(d);
}

int main()
{
    run<0, 0, 0>(), run<0, 1, 0>(), run<0, 2, 0>(), run<0, 0, 1>(), run<0, 1, 1>(), run<0, 2, 1>(), run<0, 1, 2>();
    run<1, 0, 0>(), run<1, 2, 0>(), run<1, 4, 0>(), run<1, 0, 2>(), run<1, 2, 2>(), run<1, 4, 2>(), run<1, 2, 4>();
    return 0;
}
