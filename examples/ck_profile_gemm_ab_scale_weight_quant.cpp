// What a maintainer adds to CK's ckProfiler to sweep the gfx950 block-scaled fp8 x int4 operators: CK's OWN profiling loop
// (profiler/include/profiler/profile_gemm_ab_scale_impl.hpp, used unmodified: input and scale generation, the host check on
// operands scaled to fp32, check_err, the rotating-buffer StreamConfig, Perf / Best Perf lines) instantiated for
// <fp8 A, ck::int4_t B, 128 x 128 x 128 blocks>, with the factory header of this engine included next to CK's.  Argument
// positions are those of profiler/src/profile_gemm_ab_scale.cpp:41-95; data types 8 - 11 extend its GemmDataType list.
// (CK's loop verifies at rtol = atol = 5e-2 when A is ck::f8_t and at check_err's defaults otherwise -- 1e-3 for bf16, below one
// bf16 rounding step: data type 11, E4M3FN -> bf16, therefore reports a handful of one-step differences on non-exact inputs.)
// Builds only over a CK checkout with the int4 extension on (examples/Makefile: dropin).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "ck/library/tensor_operation_instance/gpu/gemm_ab_scale_int4_values_gfx950.hpp"
#include "profiler/profile_gemm_ab_scale_impl.hpp"

int main(int argc, char* argv[])
{
    if(argc != 15 && argc != 18)
    {
        printf("arg1: tensor operation (gemm_ab_scale)\n"
               "arg2: data type (8: f8 (E4M3 FNUZ) x int4 -> f16; 9: -> bf16; 10: f8 (E4M3FN) x int4 -> f16; 11: -> bf16)\n"
               "arg3: matrix layout (1: A[m, k] * B[n, k] = C[m, n])\n"
               "arg4: scale block tile (0: ScaleBlockM/N/K = [128, 128, 128])\n"
               "arg5: verification (0: no; 1: yes)\n"
               "arg6: initialization (0: no init; 1: integer value; 2: decimal value)\n"
               "arg7: print tensor value (0: no; 1: yes)\n"
               "arg8: time kernel (0=no, 1=yes)\n"
               "arg9 to 14: M, N, K, StrideA, StrideB, StrideE\n"
               "optional: arg15 warm-up cycles (default 1), arg16 iterations (default 10), arg17 memory for the rotating buffer in MB\n");
        return 1;
    }
    const int data_type = std::stoi(argv[2]), layout = std::stoi(argv[3]), tile = std::stoi(argv[4]);
    const bool verify = std::stoi(argv[5]), log = std::stoi(argv[7]), time = std::stoi(argv[8]);
    const int init = std::stoi(argv[6]);
    const int M = std::stoi(argv[9]), N = std::stoi(argv[10]), K = std::stoi(argv[11]);
    const int sa = std::stoi(argv[12]), sb = std::stoi(argv[13]), se = std::stoi(argv[14]);
    int n_warmup = 1, n_iter = 10;
    uint64_t rotating = 0;
    if(argc == 18)
    {
        n_warmup = std::stoi(argv[15]);
        n_iter   = std::stoi(argv[16]);
        rotating = std::stoull(argv[17]) * 1024 * 1024;
    }
    using Row = ck::tensor_layout::gemm::RowMajor;
    using Col = ck::tensor_layout::gemm::ColumnMajor;
    if(layout != 1 || tile != 0)
    {
        printf("this data_type & layout is not implemented\n");
        return 1;
    }
    auto run = [&](auto a_type, auto e_type) {
        using A = decltype(a_type);
        using E = decltype(e_type);
        return ck::profiler::profile_gemm_ab_scale_impl<A, float, ck::int4_t, float, float, float, E, 128, 128, 128, Row, Col, Row>(
            verify, init, log, time, M, N, K, sa < 0 ? K : sa, sb < 0 ? K : sb, se < 0 ? N : se, n_warmup, n_iter, rotating);
    };
    bool pass = false;
    switch(data_type)
    {
    case 8: pass = run(ck::f8_t{}, ck::half_t{}); break;
    case 9: pass = run(ck::f8_t{}, ck::bhalf_t{}); break;
    case 10: pass = run(ck::wq_f8_e4m3fn_t{}, ck::half_t{}); break;
    case 11: pass = run(ck::wq_f8_e4m3fn_t{}, ck::bhalf_t{}); break;
    default: printf("this data_type & layout is not implemented\n"); return 1;
    }
    return pass ? 0 : 1;
}
