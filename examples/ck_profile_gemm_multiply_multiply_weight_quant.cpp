// What a maintainer adds to CK's ckProfiler to sweep the gfx950 fp8 x int4 MultiplyMultiply operator: CK's OWN
// profiling loop (profiler/include/profiler/profile_gemm_multiply_multiply_impl.hpp, used unmodified: input generation,
// ReferenceGemm + the MultiplyMultiply functor on the host, check_err, the rotating-buffer StreamConfig, Perf / Best
// Perf lines) instantiated for <fp8 A, ck::int4_t B>, with the factory header of this engine included next to CK's.
// Argument positions are those of profiler/src/profile_gemm_multiply_multiply.cpp:37-84 (which reads the three optional
// arguments only when argc == 18 although it admits 19; this driver reads them when they are there); data types 8 - 11
// extend its GemmDataType list.  Builds only over a CK checkout with the int4 extension on (examples/Makefile: dropin).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "ck/library/tensor_operation_instance/gpu/gemm_multiply_multiply_int4_values_gfx950.hpp"
#include "profiler/profile_gemm_multiply_multiply_impl.hpp"

int main(int argc, char* argv[])
{
    if(argc != 16 && argc != 19)
    {
        printf("arg1: tensor operation (gemm_multiply_multiply)\n"
               "arg2: data type (8: f8 (E4M3 FNUZ) x int4 -> f16; 9: -> bf16; 10: f8 (E4M3FN) x int4 -> f16; 11: -> bf16)\n"
               "arg3: matrix layout (1: A[m, k] * B[n, k] = C[m, n])\n"
               "arg4: verification (0: no; 1: yes)\n"
               "arg5: initialization (0: no init; 1: integer value; 2: decimal value)\n"
               "arg6: print tensor value (0: no; 1: yes)\n"
               "arg7: time kernel (0=no, 1=yes)\n"
               "arg8 to 15: M, N, K, StrideA, StrideB, StrideD0, StrideD1, StrideE\n"
               "optional: arg16 warm-up cycles (default 1), arg17 iterations (default 10), arg18 memory for the rotating buffer in MB\n");
        return 1;
    }
    const int data_type = std::stoi(argv[2]), layout = std::stoi(argv[3]);
    const bool verify = std::stoi(argv[4]), log = std::stoi(argv[6]), time = std::stoi(argv[7]);
    const int init = std::stoi(argv[5]);
    const int M = std::stoi(argv[8]), N = std::stoi(argv[9]), K = std::stoi(argv[10]);
    const int sa = std::stoi(argv[11]), sb = std::stoi(argv[12]), sd0 = std::stoi(argv[13]), sd1 = std::stoi(argv[14]),
              se = std::stoi(argv[15]);
    int n_warmup = 1, n_iter = 10;
    uint64_t rotating = 0;
    if(argc == 19)
    {
        n_warmup = std::stoi(argv[16]);
        n_iter   = std::stoi(argv[17]);
        rotating = std::stoull(argv[18]) * 1024 * 1024;
    }
    using Row = ck::tensor_layout::gemm::RowMajor;
    using Col = ck::tensor_layout::gemm::ColumnMajor;
    if(layout != 1)
    {
        printf("this data_type & layout is not implemented\n");
        return 1;
    }
    // (ComputeDataType float: ReferenceGemm multiplies the exact values of both operands)
    auto run = [&](auto a_type, auto e_type) {
        using A = decltype(a_type);
        using E = decltype(e_type);
        return ck::profiler::profile_gemm_multiply_multiply_impl<A, ck::int4_t, float, float, float, float, E, Row, Col, Row, Col, Row>(
            verify, init, log, time, M, N, K, sa < 0 ? K : sa, sb < 0 ? K : sb, sd0 < 0 ? N : sd0, sd1 < 0 ? M : sd1,
            se < 0 ? N : se, n_warmup, n_iter, rotating);
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
