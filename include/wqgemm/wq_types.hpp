// Storage tags of the gfx950 weight-quantized GEMM engine that have no counterpart in Composable
// Kernel's type list (include/ck/utility/data_type.hpp).  Usable from a CK build and from the
// stand-alone facade alike: nothing here depends on either.
#pragma once

#include <cstdint>

namespace ck {

// fp8 E4M3FN activations ("OCP" E4M3, the E4M3 the gfx950 matrix core decodes): one byte per
// element, bias 7, S.1111.111 = NaN, no infinities.  The reference's f8_t is E4M3 FNUZ (bias 8).
struct wq_f8_e4m3fn_t
{
    uint8_t data;
};

} // namespace ck
