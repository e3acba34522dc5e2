"""Helpers for the GPU parity tests: torch owns the device memory, the C ABI does the work."""
from __future__ import annotations

import numpy as np
import torch

from composable_kernel_amd import capi


def dev(x: np.ndarray) -> torch.Tensor:
    """numpy (uint16 / uint8 / int8 / float32 / int32) -> device tensor of the same bytes."""
    x = np.ascontiguousarray(x)
    if x.dtype == np.uint16:
        return torch.from_numpy(x.view(np.int16)).cuda()
    return torch.from_numpy(x).cuda()


def host_u16(t: torch.Tensor) -> np.ndarray:
    return t.cpu().numpy().view(np.uint16)


def f16_bits(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float16).view(np.uint16)


def bf16_bits(x) -> np.ndarray:
    return (np.asarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def bits_to_f32(bits: np.ndarray, dtype: int) -> np.ndarray:
    if dtype == capi.BF16:
        return (bits.astype(np.uint32) << 16).view(np.float32)
    return bits.view(np.float16).astype(np.float32)


def run_fpaintb(a_bits, b_dev_bytes, scale_bits, M, N, K, *, a_dtype=capi.F16, b_dtype=capi.I4_PACKED,
                lda=None, ldb=None, ldc=None, epilogue=capi.EPI_PASSTHROUGH, d0=None, d1=None,
                k_batch=1, expect_kernel=None, b_layout=capi.COL_MAJOR, flags=0, use_workspace=True, ldd0=0, ldd1=0,
                scale_group_k=0, ld_scale=0, a_block_scale=None, b_block_scale=None, scale_block=(128, 128, 128), ld_a_scale=0,
                ld_b_scale=0, c_dtype=None):
    """Run one fpAintB GEMM through the C ABI; returns the (M, ldc) uint16 output bits."""
    a_t = dev(a_bits)
    b_t = b_dev_bytes if isinstance(b_dev_bytes, torch.Tensor) else dev(b_dev_bytes)
    s_t = dev(scale_bits) if scale_bits is not None else None
    ldc = ldc or N
    # poison, with guard rows after the M real ones: a kernel that stores rows of a partial last tile is caught
    guard = 64 if M % 256 else 0
    c_t = torch.full((M + guard, ldc), 0x7BFF, dtype=torch.int16, device="cuda")
    keep = [a_t, b_t, s_t, c_t]
    d0_t = dev(d0) if d0 is not None else None
    d1_t = dev(d1) if d1 is not None else None
    keep += [d0_t, d1_t]
    sa_t = dev(np.ascontiguousarray(a_block_scale, np.float32)) if a_block_scale is not None else None
    sb_t = dev(np.ascontiguousarray(b_block_scale, np.float32)) if b_block_scale is not None else None
    keep += [sa_t, sb_t]
    p = capi.fpAintB_problem(
        a_t.data_ptr(), b_t.data_ptr(), s_t.data_ptr() if s_t is not None else None, c_t.data_ptr(), M, N, K,
        a_dtype=a_dtype, b_dtype=b_dtype, lda=lda, ldb=ldb, ldc=ldc, epilogue=epilogue,
        d0=d0_t.data_ptr() if d0_t is not None else None, d1=d1_t.data_ptr() if d1_t is not None else None,
        k_batch=k_batch, b_layout=b_layout, flags=flags, ldd0=ldd0, ldd1=ldd1, scale_group_k=scale_group_k,
        ld_scale=ld_scale, a_block_scale=sa_t.data_ptr() if sa_t is not None else None,
        b_block_scale=sb_t.data_ptr() if sb_t is not None else None, scale_block=scale_block, ld_a_scale=ld_a_scale,
        ld_b_scale=ld_b_scale,
        c_dtype=c_dtype if c_dtype is not None else (capi.F16 if a_dtype in (capi.F8_FNUZ, capi.F8_E4M3FN) else None))
    assert capi.is_supported(p), capi.lib.wqgemm_last_error().decode()
    if expect_kernel is not None:
        assert capi.kernel_name(p) == expect_kernel, (capi.kernel_name(p), capi.type_string(p))
    ws = None
    nbytes = capi.workspace_bytes(p)
    if nbytes and use_workspace:
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    capi.launch(p, ws.data_ptr() if ws is not None else None, stream)
    torch.cuda.synchronize()
    out = host_u16(c_t)
    assert (out[M:] == 0x7BFF).all(), "rows past M were written"
    return out[:M]


def pack_on_device(q: np.ndarray, K: int, N: int) -> torch.Tensor:
    """q: int8 (N, K) column-major weights -> packed (N, K//2) uint8 device tensor."""
    q_t = dev(q)
    out = torch.zeros((N, K // 2), dtype=torch.uint8, device="cuda")
    capi.pack_int4(q_t.data_ptr(), K, K, N, out.data_ptr(), K // 2, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


def check_err(out_f32: np.ndarray, ref_f32: np.ndarray, rtol: float, atol: float) -> None:
    """ck::utils::check_err semantics: |out - ref| <= atol + rtol * |ref| element-wise."""
    err = np.abs(out_f32.astype(np.float64) - ref_f32.astype(np.float64))
    lim = atol + rtol * np.abs(ref_f32.astype(np.float64))
    bad = err > lim
    assert not bad.any(), (f"{bad.sum()} of {bad.size} elements outside tolerance; "
                           f"max err {err.max():.4g} at {np.unravel_index(err.argmax(), err.shape)}")
