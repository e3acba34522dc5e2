// Path-compatible forwarding header of the gfx950 weight-quantized GEMM engine.
#pragma once
#include "wqgemm/ck_facade.hpp"
