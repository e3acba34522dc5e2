// The shapes of CK's own split-K test (test/gemm_split_k/test_gemm_splitk_ut_cases.inc:3-217 -- SmallM 1 .. 6 and MidLargeM
// {127, 255, 312, 799, 1573} at N = 512, K = 320; PaddK at K = 437; Regular 512^3 -- each with KBatch {1, 2, 3, 5, 8},
// test_gemm_splitk_util.hpp:36-87, for the four layouts mk_kn, mk_nk, km_kn, km_nk) driven through CK's OWN profiling
// loop (profiler/include/profiler/profile_gemm_splitk_impl.hpp, unmodified: GeneratorTensor inputs, ReferenceGemm,
// check_err) over the gfx950 split-K registration for 16-bit activations x int8 weights.  As in CK's fixture
// (verify on, init_method 1, no timing) a case passes when every instance that accepts it verifies; an instance may refuse
// a case ("does not support this problem", counted and printed) but never return a wrong answer.
// Builds only over a CK checkout (examples/Makefile: dropin).
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "ck/library/tensor_operation_instance/gpu/gemm_splitk_weight_quant_gfx950.hpp"
#include "profiler/profile_gemm_splitk_impl.hpp"

using Row = ck::tensor_layout::gemm::RowMajor;
using Col = ck::tensor_layout::gemm::ColumnMajor;

struct Tally
{
    int cases = 0, failed = 0, refused = 0;
};

template <typename T, typename ALayout, typename BLayout>
void run_layout(const char* name, Tally& tally, bool also_k437_for_all)
{
    const std::vector<int> kbatches{1, 2, 3, 5, 8};
    struct Shape
    {
        int M, N, K;
    };
    std::vector<Shape> shapes;
    for(int M : {1, 2, 3, 4, 5, 6, 127, 255, 312, 799, 1573})
    {
        shapes.push_back({M, 512, 320});
        if(also_k437_for_all)
            shapes.push_back({M, 512, 437});
    }
    for(int M : {15, 31, 63, 127}) // PaddK
        shapes.push_back({M, 512, 437});
    shapes.push_back({512, 512, 512}); // Regular
    for(const Shape& s : shapes)
        for(int kb : kbatches)
        {
            const int sa = std::is_same_v<ALayout, Row> ? s.K : s.M, sb = std::is_same_v<BLayout, Row> ? s.N : s.K;
            std::ostringstream captured;
            std::streambuf* old = std::cout.rdbuf(captured.rdbuf());
            const bool pass     = ck::profiler::profile_gemm_splitk_impl<T, int8_t, float, T, ALayout, BLayout, Row>(
                true, 1, false, false, s.M, s.N, s.K, sa, sb, s.N, kb, 1, 1);
            std::cout.rdbuf(old);
            const bool refused = captured.str().find("does not support this problem") != std::string::npos;
            ++tally.cases;
            tally.failed += pass ? 0 : 1;
            tally.refused += refused ? 1 : 0;
            if(!pass || refused)
                printf("%s %s M=%d N=%d K=%d KBatch=%d: %s\n", name, sizeof(T) == 2 && std::is_same_v<T, ck::half_t> ? "f16" : "bf16",
                       s.M, s.N, s.K, kb, !pass ? "WRONG RESULT" : "refused");
        }
}

int main(int argc, char** argv)
{
    const bool more = argc > 1 && std::string(argv[1]) == "all"; // K = 437 at every M as well
    Tally t;
    run_layout<ck::bhalf_t, Row, Row>("mk_kn", t, more);
    run_layout<ck::bhalf_t, Row, Col>("mk_nk", t, more);
    run_layout<ck::bhalf_t, Col, Row>("km_kn", t, more);
    run_layout<ck::bhalf_t, Col, Col>("km_nk", t, more);
    run_layout<ck::half_t, Row, Row>("mk_kn", t, more);
    run_layout<ck::half_t, Row, Col>("mk_nk", t, more);
    run_layout<ck::half_t, Col, Row>("km_kn", t, more);
    run_layout<ck::half_t, Col, Col>("km_nk", t, more);
    printf("split-K reference shapes: %d cases, %d verified, %d refused by IsSupportedArgument, %d wrong\n", t.cases,
           t.cases - t.refused - t.failed, t.refused, t.failed);
    return t.failed == 0 ? 0 : 1;
}
