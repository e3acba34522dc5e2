// What a maintainer adds to CK's ckProfiler to sweep the gfx950 split-K operator over int8 weights: CK's OWN
// profiling loop (profiler/include/profiler/profile_gemm_splitk_impl.hpp, used unmodified: input generation,
// ReferenceGemm, the KBatch list {1,2,4,8,12,16,19,20,32,38} when KBatch <= 0, check_err, Perf / Best Perf lines)
// instantiated for <16-bit float A, int8 B>, with the factory header of this engine included next to CK's.
// The argument positions are those of profiler/src/profile_gemm_splitk.cpp:36-76; data types 7 and 8 extend its
// GemmDataType list.  Builds only over a CK checkout (examples/Makefile: dropin).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "ck/library/tensor_operation_instance/gpu/gemm_splitk_weight_quant_gfx950.hpp"
#include "profiler/profile_gemm_splitk_impl.hpp"

int main(int argc, char* argv[])
{
    if(argc != 15 && argc != 17)
    {
        printf("arg1: tensor operation (gemm_splitk)\n"
               "arg2: data type (7: bf16 x int8 -> bf16; 8: fp16 x int8 -> fp16)\n"
               "arg3: matrix layout (0: A[m, k] * B[k, n] = C[m, n]; 1: A[m, k] * B[n, k] = C[m, n])\n"
               "arg4: verification (0: no; 1: yes)\n"
               "arg5: initialization (0: no init; 1: integer value; 2: decimal value)\n"
               "arg6: print tensor value (0: no; 1: yes)\n"
               "arg7: time kernel (0=no, 1=yes)\n"
               "arg8 to 13: M, N, K, StrideA, StrideB, StrideC\n"
               "arg14: split k into mulitiple batch\n"
               "optional: arg15 warm-up cycles (default 1), arg16 iterations (default 10)\n");
        return 1;
    }
    const int data_type = std::stoi(argv[2]), layout = std::stoi(argv[3]);
    const bool verify = std::stoi(argv[4]), log = std::stoi(argv[6]), time = std::stoi(argv[7]);
    const int init = std::stoi(argv[5]);
    const int M = std::stoi(argv[8]), N = std::stoi(argv[9]), K = std::stoi(argv[10]);
    int sa = std::stoi(argv[11]), sb = std::stoi(argv[12]), sc = std::stoi(argv[13]);
    const int kbatch = std::stoi(argv[14]);
    const int n_warmup = argc == 17 ? std::stoi(argv[15]) : 1, n_iter = argc == 17 ? std::stoi(argv[16]) : 10;
    using Row = ck::tensor_layout::gemm::RowMajor;
    using Col = ck::tensor_layout::gemm::ColumnMajor;
    sa = sa < 0 ? K : sa;
    sc = sc < 0 ? N : sc;
    auto run = [&](auto a_type, auto b_layout) {
        using T       = decltype(a_type);
        using BLayout = decltype(b_layout);
        const int sb_ = sb < 0 ? (ck::is_same_v<BLayout, Row> ? N : K) : sb;
        return ck::profiler::profile_gemm_splitk_impl<T, int8_t, float, T, Row, BLayout, Row>(
            verify, init, log, time, M, N, K, sa, sb_, sc, kbatch, n_warmup, n_iter);
    };
    bool pass = false;
    if(data_type == 7 && layout == 0)
        pass = run(ck::bhalf_t{}, Row{});
    else if(data_type == 7 && layout == 1)
        pass = run(ck::bhalf_t{}, Col{});
    else if(data_type == 8 && layout == 0)
        pass = run(ck::half_t{}, Row{});
    else if(data_type == 8 && layout == 1)
        pass = run(ck::half_t{}, Col{});
    else
    {
        printf("this data_type & layout is not implemented\n");
        return 1;
    }
    return pass ? 0 : 1;
}
