// Storage tags of the gfx950 weight-quantized GEMM engine that have no counterpart in Composable
// Kernel's type list (include/ck/utility/data_type.hpp).  Usable from a CK build and from the
// stand-alone facade alike: nothing here depends on either.
#pragma once

#include <cmath>
#include <cstdint>

namespace ck {

// fp8 E4M3FN activations ("OCP" E4M3, the E4M3 the gfx950 matrix core decodes): one byte per
// element, bias 7, S.1111.111 = NaN, no infinities.  The reference's f8_t is E4M3 FNUZ (bias 8).
// The conversions make the type usable where CK's generic host code handles element types by static_cast
// (type_convert's primary template, include/ck/utility/type_convert.hpp:17-25; the tensor generators of
// library/include/ck/library/utility/host_tensor_generator.hpp; ReferenceGemm): to float exactly; from float by
// round to nearest even, saturating at +-448, NaN kept.
struct wq_f8_e4m3fn_t
{
    uint8_t data;

    wq_f8_e4m3fn_t() = default;
    explicit wq_f8_e4m3fn_t(float x) : data(from_float(x)) {}
    explicit wq_f8_e4m3fn_t(double x) : data(from_float((float)x)) {}
    explicit wq_f8_e4m3fn_t(int x) : data(from_float((float)x)) {}

    explicit operator float() const
    {
        if((data & 0x7F) == 0x7F)
            return std::nanf("");
        const int e   = (data >> 3) & 15, m = data & 7;
        const float v = e == 0 ? (float)m * 0.001953125f /* m * 2^-9 */ : std::ldexp((float)(8 + m), e - 10);
        return (data & 0x80) ? -v : v;
    }

    static uint8_t from_float(float x)
    {
        if(std::isnan(x))
            return 0x7F;
        const uint8_t sign = std::signbit(x) ? 0x80 : 0x00;
        float a            = std::fabs(x);
        if(a >= 464.f) // (448 is the largest value; halfway to the next step of 32 rounds to even = up: saturate)
            return sign | 0x7E;
        int e;
        (void)std::frexp(a, &e); // a = f * 2^e, f in [0.5, 1)
        int exp = e - 1;         // a = 1.xxx * 2^exp
        if(exp < -6)
            exp = -6; // subnormal range: steps of 2^-9
        const float step = std::ldexp(1.f, exp - 3);
        const float q    = std::nearbyint(a / step); // round to nearest even (default rounding mode)
        const float r    = q * step;
        if(r == 0.f)
            return sign;
        if(r > 448.f)
            return sign | 0x7E;
        int e2;
        (void)std::frexp(r, &e2);
        int exp2 = e2 - 1;
        if(exp2 < -6)
            return (uint8_t)(sign | (uint8_t)(r / 0.001953125f));
        const int mant = (int)(r / std::ldexp(1.f, exp2 - 3)) - 8;
        return (uint8_t)(sign | ((exp2 + 7) << 3) | mant);
    }
};

} // namespace ck
