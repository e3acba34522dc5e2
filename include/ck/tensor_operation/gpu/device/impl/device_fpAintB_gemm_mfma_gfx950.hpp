// The fpAintB operator of the gfx950 engine; see device_wq_gemm_gfx950.hpp.
#pragma once
#include "ck/tensor_operation/gpu/device/impl/device_wq_gemm_gfx950.hpp"
