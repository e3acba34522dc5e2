"""Offline weight preparation for the weight-quantized GEMM engine: per-channel or group-wise symmetric
quantisation, packing and a small on-disk container.

The reference keeps quantised weights as one int8 (or int4-valued int8) per element and re-encodes them on the
host right before the GEMM (example/64_fpAintB_gemm/common.hpp:51-83, UnsignedWeightPreprocessor).  Here that step
is done once, offline: weights are stored K-contiguous per output channel in the engine's device formats
(``WQ_I4_PACKED``: two offset-binary nibbles per byte in the order documented at ``wqgemm_pack_int4``;
``WQ_I8``: plain int8), next to their per-channel scales.

Group-wise scales (one scale per output channel and per ``group_k`` consecutive k) are stored as a (K / group_k, N)
array, one row of N scales per group: the (K, N) scale tensor of the reference's fpAintB GEMM
(library/include/ck/library/reference_tensor_operation/cpu/reference_fpAintB_gemm.hpp:62-127) with repeated rows
collapsed, in the (K / block, N) shape of example/65_gemm_multiply_multiply/gemm_multiply_multiply_xdl_fp8_ab_scale.cpp:126-152.

Weights can additionally be stored pre-shuffled (``preshuffle``; ``WQ_I4_PRESHUFFLED`` / ``WQ_I8_PRESHUFFLED``, the layouts
documented at ``wqgemm_preshuffle_int4`` / ``wqgemm_preshuffle_int8``): the same bytes with their 16-byte units (32 or 16
consecutive k of one column) in the order the
tile kernels fetch them, so that every wave-wide weight fetch is whole 128-byte lines.  Needs N a multiple of 32 and K
of 64; every matrix-core kernel reads it.

Container (``save_packed`` / ``load_packed``): a NumPy ``.npz`` archive with the arrays ``weight`` (uint8 or int8,
shape (N, K/2) or (N, K); for the pre-shuffled layout the N * K / 2 bytes keep the (N, K/2) shape although rows no longer
are columns), ``scale`` (uint16 bit patterns of fp16 / bf16, shape (N,), or (K / group_k, N) for
group-wise scales) and the 0-d entries ``format`` ("wqgemm-weights-3"; "wqgemm-weights-2" files, which have no ``layout``,
and "wqgemm-weights-1" files, which have no ``group_k`` either, still load), ``bits`` (4 or 8), ``scale_dtype`` ("f16" or
"bf16"), ``N``, ``K``, ``group_k`` (0 = one scale per channel) and ``layout`` ("packed" or "preshuffled").  Loaded with
``allow_pickle=False``.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import capi

FORMAT = "wqgemm-weights-3"
FORMATS_READ = ("wqgemm-weights-1", "wqgemm-weights-2", FORMAT)
LAYOUTS = ("packed", "preshuffled")


@dataclass
class PackedWeights:
    weight: np.ndarray      # uint8 (N, K//2) for 4 bits, int8 (N, K) for 8 bits; K contiguous
    scale: np.ndarray       # uint16 bit patterns, (N,) or (K // group_k, N)
    bits: int
    scale_dtype: str        # "f16" | "bf16"
    N: int
    K: int
    group_k: int = 0        # 0: one scale per output channel; g: one per channel and per g consecutive k
    layout: str = "packed"  # "preshuffled": 4-bit weights in the order of wqgemm_preshuffle_int4 (K pitch = K)

    @property
    def b_dtype(self) -> int:
        if self.bits == 4:
            return capi.I4_PRESHUFFLED if self.layout == "preshuffled" else capi.I4_PACKED
        return capi.I8_PRESHUFFLED if self.layout == "preshuffled" else capi.I8

    @property
    def a_dtype(self) -> int:
        return capi.F16 if self.scale_dtype == "f16" else capi.BF16


def _to_bits(x: np.ndarray, dtype: str) -> np.ndarray:
    x = np.asarray(x, np.float32)
    if dtype == "f16":
        return x.astype(np.float16).view(np.uint16)
    return (x.view(np.uint32) >> 16).astype(np.uint16)  # bf16 by truncation (include/ck/utility/type_convert.hpp:53-64)


def _from_bits(b: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "f16":
        return b.view(np.float16).astype(np.float32)
    return (b.astype(np.uint32) << 16).view(np.float32)


def quantize_per_channel(w: np.ndarray, bits: int = 4, scale_dtype: str = "f16") -> tuple[np.ndarray, np.ndarray]:
    """Symmetric round-to-nearest quantisation of w (N, K), one scale per output channel n.

    Returns (q int8 (N, K) in [-2^(bits-1), 2^(bits-1)-1], scale bit patterns uint16 (N,)).  The scale is
    amax / (2^(bits-1) - 1) rounded to the 16-bit type FIRST, and q is computed against the rounded scale, so
    dequantising with the stored scale reproduces the grid the quantiser used."""
    if bits not in (4, 8):
        raise ValueError("bits must be 4 or 8")
    w = np.asarray(w, np.float32)
    qmax = 2 ** (bits - 1) - 1
    amax = np.abs(w).max(axis=1)
    s_bits = _to_bits(np.where(amax > 0, amax / qmax, 1.0), scale_dtype)
    s = _from_bits(s_bits, scale_dtype)
    q = np.clip(np.rint(w / s[:, None]), -qmax - 1, qmax).astype(np.int8)
    return q, s_bits


def quantize_per_group(w: np.ndarray, group_k: int, bits: int = 4,
                       scale_dtype: str = "f16") -> tuple[np.ndarray, np.ndarray]:
    """quantize_per_channel applied to every group of ``group_k`` consecutive k of every output channel.

    Returns (q int8 (N, K), scale bit patterns uint16 (K // group_k, N)).  K must be a multiple of group_k."""
    if bits not in (4, 8):
        raise ValueError("bits must be 4 or 8")
    w = np.asarray(w, np.float32)
    N, K = w.shape
    if group_k <= 0 or K % group_k:
        raise ValueError("K must be a multiple of group_k")
    qmax = 2 ** (bits - 1) - 1
    wg = w.reshape(N, K // group_k, group_k)
    amax = np.abs(wg).max(axis=2)                                   # (N, groups)
    s_bits = _to_bits(np.where(amax > 0, amax / qmax, 1.0), scale_dtype)
    s = _from_bits(s_bits, scale_dtype)
    q = np.clip(np.rint(wg / s[:, :, None]), -qmax - 1, qmax).astype(np.int8).reshape(N, K)
    return q, np.ascontiguousarray(s_bits.T)


def pack(q: np.ndarray, scale_bits: np.ndarray, bits: int = 4, scale_dtype: str = "f16", group_k: int = 0) -> PackedWeights:
    """q int8 (N, K) -> the engine's device format; scale_bits (N,), or (K // group_k, N) with group_k > 0."""
    q = np.ascontiguousarray(q, np.int8)
    N, K = q.shape
    scale_bits = np.ascontiguousarray(scale_bits, np.uint16)
    if scale_bits.shape != ((K // group_k, N) if group_k else (N,)) or (group_k and K % group_k):
        raise ValueError("scale shape does not match N, K and group_k")
    if bits == 4:
        if K % 8:
            raise ValueError("K must be a multiple of 8 for packed int4")
        weight = capi.pack_int4_host(q, K, N)
    else:
        weight = q
    return PackedWeights(weight, scale_bits, bits, scale_dtype, N, K, group_k)


def _preshuffled_unit_order(N: int, K: int, bits: int) -> np.ndarray:
    """dest[n, j] = index (in 16-byte units) at which the pre-shuffled layout keeps unit j of column n (4 bits: unit j =
    k from 32 j, blocks of 32 columns; 8 bits: k from 16 j, blocks of 16 columns)."""
    n = np.arange(N)[:, None]
    if bits == 4:
        j = np.arange(K // 32)[None, :]
        return (n // 32) * K + (j // 2) * 64 + 32 * ((n % 32) // 16) + 16 * (j % 2) + n % 16
    j = np.arange(K // 16)[None, :]
    return (n // 16) * K + (j // 4) * 64 + 4 * (n % 16) + j % 4


def preshuffle(pw: PackedWeights) -> PackedWeights:
    """Packed int4 or int8 weights -> the pre-shuffled layout (through the library's host routines)."""
    if pw.layout != "packed":
        raise ValueError("preshuffle takes weights in the packed layout")
    if pw.N % 32 or pw.K % 64:
        raise ValueError("the pre-shuffled layout needs N % 32 == 0 and K % 64 == 0")
    if pw.bits == 4:
        w = capi.preshuffle_int4_host(pw.weight, pw.K, pw.N).reshape(pw.N, pw.K // 2)
    else:
        w = capi.preshuffle_int8_host(pw.weight, pw.K, pw.N).reshape(pw.N, pw.K)
    return PackedWeights(w, pw.scale, pw.bits, pw.scale_dtype, pw.N, pw.K, pw.group_k, "preshuffled")


def unshuffle(pw: PackedWeights) -> PackedWeights:
    """Inverse of ``preshuffle`` (host side, for inspection and tests; the index map is restated here in numpy)."""
    if pw.layout != "preshuffled":
        return pw
    units = np.ascontiguousarray(pw.weight).reshape(-1, 16)
    w = units[_preshuffled_unit_order(pw.N, pw.K, pw.bits)].reshape(pw.weight.shape)
    return PackedWeights(np.ascontiguousarray(w), pw.scale, pw.bits, pw.scale_dtype, pw.N, pw.K, pw.group_k, "packed")


def unpack_int4(packed: np.ndarray, K: int) -> np.ndarray:
    """Inverse of the packed int4 layout (host side, for inspection and tests): uint8 (N, K//2) -> int8 (N, K).
    Inside each little-endian 32-bit word nibble p holds k = 8g + 2p (p < 4) or 8g + 2(p - 4) + 1 (p >= 4)."""
    words = np.ascontiguousarray(packed).view("<u4")            # (N, K//8)
    N = words.shape[0]
    out = np.empty((N, K // 8, 8), np.int8)
    for p in range(8):
        k = 2 * p if p < 4 else 2 * (p - 4) + 1
        out[:, :, k] = ((words >> (4 * p)) & 0xF).astype(np.int16) - 8
    return out.reshape(N, K)


def dequantize(pw: PackedWeights) -> np.ndarray:
    """fp32 (N, K) weights as the kernels see them: q * scale rounded to the scale's 16-bit type."""
    pw = unshuffle(pw)
    q = unpack_int4(pw.weight, pw.K) if pw.bits == 4 else pw.weight
    s = _from_bits(pw.scale, pw.scale_dtype)
    s_nk = np.repeat(s.T, pw.group_k, axis=1) if pw.group_k else s[:, None]
    return _from_bits(_to_bits(q.astype(np.float32) * s_nk, pw.scale_dtype), pw.scale_dtype)


def save_packed(path: str, pw: PackedWeights) -> None:
    np.savez(path, format=np.array(FORMAT), bits=np.array(pw.bits), scale_dtype=np.array(pw.scale_dtype),
             N=np.array(pw.N), K=np.array(pw.K), group_k=np.array(pw.group_k), layout=np.array(pw.layout), weight=pw.weight,
             scale=pw.scale)


def load_packed(path: str) -> PackedWeights:
    with np.load(path, allow_pickle=False) as z:
        if str(z["format"]) not in FORMATS_READ:
            raise ValueError(f"{path}: not a {FORMAT} container")
        pw = PackedWeights(z["weight"], z["scale"], int(z["bits"]), str(z["scale_dtype"]), int(z["N"]), int(z["K"]),
                           int(z["group_k"]) if "group_k" in z.files else 0,
                           str(z["layout"]) if "layout" in z.files else "packed")
    if pw.layout not in LAYOUTS or (pw.layout == "preshuffled" and (pw.N % 32 or pw.K % 64)):
        raise ValueError(f"{path}: layout {pw.layout!r} does not go with bits = {pw.bits}, N = {pw.N}, K = {pw.K}")
    want = (pw.N, pw.K // 2) if pw.bits == 4 else (pw.N, pw.K)
    want_s = (pw.K // pw.group_k, pw.N) if pw.group_k > 0 else (pw.N,)
    if pw.group_k < 0 or (pw.group_k and pw.K % pw.group_k) or pw.weight.shape != want or pw.scale.shape != want_s:
        raise ValueError(f"{path}: array shapes do not match N = {pw.N}, K = {pw.K}, bits = {pw.bits}, group_k = {pw.group_k}")
    return pw
