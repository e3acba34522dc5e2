// Issue rates of the vector instructions the tile epilogues are made of, no MFMA beside them: cycles per instruction per
// SIMD (s_memtime), one and two waves per SIMD, independent operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(512, 2) void probe(long long* out, int iters, float seed)
{
    typedef float f2_t __attribute__((ext_vector_type(2)));
    float x[16];
    f2_t v[8];
    for(int i = 0; i < 16; ++i)
        x[i] = seed + threadIdx.x * 0.001f + i;
    for(int i = 0; i < 8; ++i)
        v[i] = f2_t{x[2 * i], x[2 * i + 1]};
    const f2_t c = {1.0001f, 0.9999f};
    const float cs = 1.0001f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for(int it = 0; it < iters; ++it)
    {
#pragma unroll
        for(int r = 0; r < 4; ++r)
#pragma unroll
            for(int i = 0; i < 8; ++i)
            {
                if constexpr(KIND == 0)
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
                else if constexpr(KIND == 1)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
                else if constexpr(KIND == 2)
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(cs));
                else if constexpr(KIND == 3)
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(cs));
                else if constexpr(KIND == 4)
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                else if constexpr(KIND == 5)
                    asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
                else if constexpr(KIND == 6)
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
                else if constexpr(KIND == 7)
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
                else if constexpr(KIND == 8)
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
                else if constexpr(KIND == 9)
                    asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x[i]));
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for(int i = 0; i < 16; ++i)
        s += x[i];
    for(int i = 0; i < 8; ++i)
        s += v[i][0] + v[i][1];
    if(threadIdx.x % 64 == 0)
        out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
    if(s == 1.2345f)
        out[0] = 0;
}

template <int KIND>
void run(const char* name)
{
    long long* d;
    hipMalloc(&d, 256 * 8 * 8);
    for(int threads : {256, 512})
    {
        const int iters = 2000;
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 0, 0, d, 100, 1.f);
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 0, 0, d, iters, 1.f);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * 8);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        const int waves = threads / 64;
        for(int b = 0; b < 256; ++b)
            for(int w = 0; w < waves; ++w)
                sum += (double)h[b * 8 + w];
        const double per_wave = sum / (256 * waves) / (iters * 32.0);
        printf("%-22s waves/SIMD=%d: %5.2f cycles per instruction per wave, %5.2f per SIMD\n", name, waves / 4, per_wave, per_wave / (waves / 4));
    }
    hipFree(d);
}

int main()
{
    run<0>("v_pk_mul_f32");
    run<8>("v_pk_add_f32");
    run<1>("v_pk_fma_f32");
    run<2>("v_mul_f32");
    run<3>("v_fma_f32");
    run<4>("v_exp_f32");
    run<5>("v_rcp_f32");
    run<6>("v_cvt_pk_bf16_f32");
    run<7>("v_cvt_pkrtz_f16_f32");
    run<9>("v_cvt_f16_f32");
    return 0;
}
