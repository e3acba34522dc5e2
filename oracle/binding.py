"""ctypes bindings for the CPU checkers (TEST INFRASTRUCTURE ONLY).

Two shared objects expose the same set of functions under different prefixes:

* ``oracle/libwq_oracle.so`` (prefix ``wqo_``)   - the plain-C restatement, built by gcc;
* ``oracle/_ref/libck_ref.so`` (prefix ``ckref_``) - the reference's own host templates, built
  only where the reference tree is mounted.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  The product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "libwq_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libck_ref.so")

_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build_port() -> str:
    """Compile the plain-C restatement if it is missing or stale."""
    src = os.path.join(HERE, "wq_oracle.c")
    if not os.path.exists(PORT_SO) or os.path.getmtime(PORT_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "port"])
    return PORT_SO


def build_ref(ck_ref: str = "/root/reference") -> str | None:
    """Compile the reference-backed checker when the reference tree is present."""
    if not os.path.isdir(ck_ref):
        return REF_SO if os.path.exists(REF_SO) else None
    src = os.path.join(HERE, "ref_shim.cpp")
    if not os.path.exists(REF_SO) or os.path.getmtime(REF_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", f"CK_REF={ck_ref}"])
    return REF_SO


class Checker:
    """Uniform view over either shared object.  ``kind`` is "port" or "reference"."""

    def __init__(self, path: str, prefix: str, kind: str):
        self.kind = kind
        self.path = path
        self._lib = C.CDLL(path)
        self._p = prefix
        self._threaded = prefix == "wqo_"  # the port takes an explicit thread count
        L, p = self._lib, prefix

        def sig(name, res, args):
            fn = getattr(L, p + name)
            fn.restype = res
            fn.argtypes = args
            return fn

        self.f32_to_f16 = sig("f32_to_f16", C.c_uint16, [C.c_float])
        self.f16_to_f32 = sig("f16_to_f32", C.c_float, [C.c_uint16])
        self.f32_to_bf16 = sig("f32_to_bf16", C.c_uint16, [C.c_float])
        self.bf16_to_f32 = sig("bf16_to_f32", C.c_float, [C.c_uint16])
        self.f8_to_f32 = sig("f8_to_f32" if p == "ckref_" else "f8_fnuz_to_f32", C.c_float, [C.c_uint8])
        self.f32_to_i8 = sig("f32_to_i8", C.c_int8, [C.c_float])
        self.dequant_f16 = sig("dequant_f16", C.c_uint16, [C.c_int8, C.c_uint16])
        self.multiply_bf16 = sig("multiply_bf16", C.c_uint16, [C.c_int8, C.c_uint16])
        self.fastgelu = sig("fastgelu", C.c_float, [C.c_float])
        self.act_mul_clamp = sig("act_mul_clamp", C.c_int8, [C.c_float, C.c_int32])
        self.act_mul_clamp_from_float = sig("act_mul_clamp_from_float", C.c_int8, [C.c_float, C.c_float])
        self.multiply_multiply_f16 = sig("multiply_multiply_f16", C.c_uint16, [C.c_float] * 3)
        self.multiply_multiply_bf16 = sig("multiply_multiply_bf16", C.c_uint16, [C.c_float] * 3)
        self.multiply_add_bf16 = sig("multiply_add_bf16", C.c_uint16, [C.c_float, C.c_uint16, C.c_uint16])
        self.bf16_convert_rtn = sig("bf16_convert_rtn", C.c_uint16, [C.c_float])
        self.multiply_add_fastgelu_bf16 = sig(
            "multiply_add_fastgelu_bf16", C.c_uint16, [C.c_float, C.c_uint16, C.c_uint16])
        self._fill = sig("fill_uniform", None, [C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_int])
        self._gen2 = sig("generator2_i8", None,
                         [_i8p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int])
        tail = [C.c_int] if self._threaded else []
        self._fpaintb = sig("fpAintB_gemm_f16", None, [_u16p, _i8p, _u16p, _u16p] + [C.c_int] * 6 + tail)
        self._fpaintb_g = sig("fpAintB_gemm_f16_grouped", None, [_u16p, _i8p, _u16p, _u16p] + [C.c_int] * 8 + tail)
        self._i8 = sig("gemm_i8_requant", None, [_i8p, _i8p, _i8p] + [C.c_int] * 6 + [C.c_float] + tail)
        self._i8b = sig("gemm_i8_bias_requant", None,
                        [_i8p, _i8p, C.c_void_p, _i8p] + [C.c_int] * 6 + [C.c_float, C.c_int] + tail)
        self._bf16 = sig("gemm_bf16_i8", None,
                         [_u16p, _i8p, C.c_void_p, _u16p] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p] + tail)
        self._bf16_g = sig("gemm_bf16_i8_grouped", None,
                           [_u16p, _i8p, C.c_void_p, _u16p] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + tail)
        self._f8 = sig("gemm_f8_i8_mm", None, [_u8p, _i8p, _f32p, _f32p, _u16p] + [C.c_int] * 7 + tail)
        # (the plain-C port takes one more int, e4m3fn, before the thread count; the reference shim knows FNUZ only)
        self._f8_ab = sig("gemm_f8_i8_abscale", None, [_u8p, _i8p, _f32p, _f32p, _u16p] + [C.c_int] * 12 +
                          ([C.c_int] if self._threaded else []) + tail)
        self._bf16_dmat = sig("gemm_bf16_i8_dmat", None,
                              [_u16p, _i8p, C.c_void_p, _u16p] + [C.c_int] * 8 +
                              [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + tail)
        if not self._threaded:
            self._bf16_unscaled = sig("gemm_bf16_i8_unscaled", None, [_u16p, _i8p, _u16p] + [C.c_int] * 7)
        if self._threaded:
            self._partial = sig("fpAintB_gemm_f16_partial", None,
                                [_u16p, _i8p, _u16p, _f32p] + [C.c_int] * 7)
            self.splitk_kread = sig("splitk_kread", C.c_int, [C.c_int] * 3)
            self._reduce = sig("splitk_reduce_f16", None, [_f32p, C.c_int, C.c_int, C.c_int, _u16p, C.c_int])
            self._pack = sig("pack_int4", C.c_int, [_i8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int])
            self._unpack = sig("unpack_int4", C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, _i8p, C.c_int])
            self._preshuffle = sig("preshuffle_int4", C.c_int, [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int])
            self._preshuffle8 = sig("preshuffle_int8", C.c_int, [_i8p, C.c_int, C.c_int, C.c_int, _i8p, C.c_int])
            self._offs = sig("offset_u8", None, [_i8p, _u8p, C.c_size_t])

    # -- generators ---------------------------------------------------------------------------
    _DT = {"f16": (0, np.uint16), "bf16": (1, np.uint16), "i8": (2, np.int8), "f32": (3, np.float32)}

    def fill_uniform(self, n: int, dtype: str, a: float, b: float, integer_value: bool) -> np.ndarray:
        code, npdt = self._DT[dtype]
        out = np.empty(n, dtype=npdt)
        self._fill(out.ctypes.data, n, code, a, b, int(integer_value))
        return out

    def generator2_i8(self, rows, cols, sr, sc, lo, hi, seed=-1) -> np.ndarray:
        space = 1 + (rows - 1) * sr + (cols - 1) * sc
        out = np.zeros(space, dtype=np.int8)
        self._gen2(out, rows, cols, sr, sc, lo, hi, seed)
        return out

    # -- GEMMs --------------------------------------------------------------------------------
    def _t(self, nthreads):
        return [int(nthreads or os.cpu_count() or 1)] if self._threaded else []

    def fpAintB_gemm_f16(self, a, b, scale, M, N, K, lda, ldb, ldc, nthreads=None) -> np.ndarray:
        """a: uint16 (M,lda) fp16 bits; b: int8 column-major (N,ldb); scale: uint16 (N)."""
        c = np.zeros(1 + (M - 1) * ldc + (N - 1), dtype=np.uint16)
        self._fpaintb(a.ravel(), b.ravel(), scale, c, M, N, K, lda, ldb, ldc, *self._t(nthreads))
        return c

    def fpAintB_gemm_f16_grouped(self, a, b, scale, M, N, K, lda, ldb, ldc, group_k, ld_scale=None,
                                 nthreads=None) -> np.ndarray:
        """Group-wise scales: scale uint16 (ceil(K / group_k), ld_scale), one row of N scales per group_k consecutive k."""
        ld_scale = ld_scale or N
        c = np.zeros(1 + (M - 1) * ldc + (N - 1), dtype=np.uint16)
        self._fpaintb_g(a.ravel(), b.ravel(), np.ascontiguousarray(scale, np.uint16).ravel(), c, M, N, K, lda, ldb, ldc,
                        group_k, ld_scale, *self._t(nthreads))
        return c

    def gemm_i8_requant(self, a, b, M, N, K, lda, ldb, lde, requant, nthreads=None) -> np.ndarray:
        e = np.zeros(1 + (M - 1) * lde + (N - 1), dtype=np.int8)
        self._i8(a.ravel(), b.ravel(), e, M, N, K, lda, ldb, lde, requant, *self._t(nthreads))
        return e

    def gemm_i8_bias_requant(self, a, b, bias, M, N, K, lda, ldb, lde, requant, relu, nthreads=None) -> np.ndarray:
        """int32-accumulated int8 GEMM + (Add_)Activation_Mul_Clamp<PassThrough|Relu>; bias = N int32 or None."""
        e = np.zeros(1 + (M - 1) * lde + (N - 1), dtype=np.int8)
        pb = np.ascontiguousarray(bias, np.int32) if bias is not None else None
        self._i8b(a.ravel(), b.ravel(), pb.ctypes.data if pb is not None else None, e, M, N, K, lda, ldb, lde,
                  requant, int(relu), *self._t(nthreads))
        return e

    def gemm_bf16_i8(self, a, b0, b1, M, N, K, lda, b_sk, b_sn, lde, epilogue, d0=None, d1=None,
                     nthreads=None) -> np.ndarray:
        e = np.zeros(1 + (M - 1) * lde + (N - 1), dtype=np.uint16)
        p0 = d0.ctypes.data if d0 is not None else None
        p1 = d1.ctypes.data if d1 is not None else None
        pb1 = np.ascontiguousarray(b1, np.uint16).ctypes.data if b1 is not None else None  # None: B = PassThrough(b0)
        self._bf16(a.ravel(), b0.ravel(), pb1, e, M, N, K, lda, b_sk, b_sn, lde, epilogue, p0, p1,
                   *self._t(nthreads))
        return e

    def gemm_bf16_i8_grouped(self, a, b0, b1, M, N, K, lda, b_sk, b_sn, lde, epilogue, group_k, ld_b1=None, d0=None,
                             d1=None, nthreads=None) -> np.ndarray:
        """gemm_bf16_i8 with group-wise scales: b1 uint16 (ceil(K / group_k), ld_b1), one row of N scales per group."""
        ld_b1 = ld_b1 or N
        e = np.zeros(1 + (M - 1) * lde + (N - 1), dtype=np.uint16)
        p0 = d0.ctypes.data if d0 is not None else None
        p1 = d1.ctypes.data if d1 is not None else None
        b1 = np.ascontiguousarray(b1, np.uint16)
        self._bf16_g(a.ravel(), b0.ravel(), b1.ctypes.data, e, M, N, K, lda, b_sk, b_sn, lde, epilogue, p0, p1, group_k, ld_b1,
                     *self._t(nthreads))
        return e

    def gemm_bf16_i8_dmat(self, a, b0, b1, M, N, K, lda, b_sk, b_sn, lde, epilogue, d0, ldd0, d1=None, ldd1=0,
                          nthreads=None) -> np.ndarray:
        """gemm_bf16_i8 with D tensors addressed as row-major (M, N) matrices of row stride ldd (0 = per column)."""
        e = np.zeros(1 + (M - 1) * lde + (N - 1), dtype=np.uint16)
        d0 = np.ascontiguousarray(d0, np.uint16)
        d1 = np.ascontiguousarray(d1, np.uint16) if d1 is not None else None
        pb1 = np.ascontiguousarray(b1, np.uint16).ctypes.data if b1 is not None else None
        self._bf16_dmat(a.ravel(), b0.ravel(), pb1, e, M, N, K, lda, b_sk, b_sn, lde, epilogue, d0.ctypes.data, ldd0,
                        d1.ctypes.data if d1 is not None else None, ldd1, *self._t(nthreads))
        return e

    def gemm_bf16_i8_unscaled(self, a, b, M, N, K, lda, b_sk, b_sn, ldc) -> np.ndarray:
        """Reference-backed checker only: ReferenceGemm<bhalf, int8, bhalf, float> straight on int8 B (example/35)."""
        c = np.zeros(1 + (M - 1) * ldc + (N - 1), dtype=np.uint16)
        self._bf16_unscaled(a.ravel(), b.ravel(), c, M, N, K, lda, b_sk, b_sn, ldc)
        return c

    def gemm_f8_i8_mm(self, a, b, d0, d1, M, N, K, lda, ldb, lde, fastgelu, nthreads=None) -> np.ndarray:
        e = np.zeros(1 + (M - 1) * lde + (N - 1), dtype=np.uint16)
        self._f8(a.ravel(), b.ravel(), d0, d1, e, M, N, K, lda, ldb, lde, int(fastgelu), *self._t(nthreads))
        return e

    def gemm_f8_i8_abscale(self, a, b, sa, sb, M, N, K, lda, ldb, lde, block, ld_sa, ld_sb, e_bf16=False, e4m3fn=False,
                           nthreads=None) -> np.ndarray:
        """fp8 A x int8-held weights with block-wise scales of both operands (example/65's fp8_ab_scale host check):
        block = (bm, bn, bk); sa row-major (M / bm, K / bk), sb column-major (K / bk, N / bn).  Returns 16-bit patterns."""
        if e4m3fn and not self._threaded:
            raise ValueError("the reference's f8_t is E4M3 FNUZ only")
        e = np.zeros(1 + (M - 1) * lde + (N - 1), dtype=np.uint16)
        extra = [int(e4m3fn)] if self._threaded else []
        self._f8_ab(a.ravel(), b.ravel(), np.ascontiguousarray(sa, np.float32).ravel(), np.ascontiguousarray(sb, np.float32).ravel(),
                    e, M, N, K, lda, ldb, lde, *block, ld_sa, ld_sb, int(e_bf16), *extra, *self._t(nthreads))
        return e

    # -- port-only helpers --------------------------------------------------------------------
    def fpAintB_partial(self, a, b, scale, M, N, lda, ldb, k_begin, k_end, nthreads=None) -> np.ndarray:
        out = np.zeros(M * N, dtype=np.float32)
        self._partial(a.ravel(), b.ravel(), scale, out, M, N, lda, ldb, k_begin, k_end,
                      int(nthreads or os.cpu_count() or 1))
        return out.reshape(M, N)

    def splitk_reduce_f16(self, partials: np.ndarray, ldc: int | None = None) -> np.ndarray:
        kb, M, N = partials.shape
        ldc = ldc or N
        c = np.zeros(1 + (M - 1) * ldc + (N - 1), dtype=np.uint16)
        self._reduce(np.ascontiguousarray(partials, dtype=np.float32).ravel(), kb, M, N, c, ldc)
        return c

    def pack_int4(self, q: np.ndarray, K: int, N: int, ldq: int | None = None) -> np.ndarray:
        """q: int8 column-major, element (k,n) at n*ldq + k.  Returns (N, K//2) uint8."""
        ldq = ldq or K
        out = np.zeros((N, K // 2), dtype=np.uint8)
        rc = self._pack(np.ascontiguousarray(q).ravel(), ldq, K, N, out.reshape(-1), K // 2)
        if rc != 0:
            raise ValueError("pack_int4: K must be a multiple of 8 and values within -8..7")
        return out

    def preshuffle_int4(self, packed: np.ndarray, K: int, N: int, ldk: int | None = None) -> np.ndarray:
        """packed: (N, K//2) uint8 in the packed int4 order -> flat uint8 in the pre-shuffled order (K pitch ldk)."""
        ldk = ldk or K
        out = np.zeros(N * ldk // 2, dtype=np.uint8)
        rc = self._preshuffle(np.ascontiguousarray(packed).reshape(-1), K // 2, K, N, out, ldk)
        if rc != 0:
            raise ValueError("preshuffle_int4: K % 64, N % 32, ldk % 64 must be 0 and ldk >= K")
        return out

    def preshuffle_int8(self, b: np.ndarray, K: int, N: int, ldk: int | None = None) -> np.ndarray:
        """b: (N, K) int8, K contiguous -> flat int8 in the pre-shuffled order (K pitch ldk)."""
        ldk = ldk or K
        out = np.zeros(N * ldk, dtype=np.int8)
        rc = self._preshuffle8(np.ascontiguousarray(b, dtype=np.int8).reshape(-1), K, K, N, out, ldk)
        if rc != 0:
            raise ValueError("preshuffle_int8: K % 64, N % 16, ldk % 64 must be 0 and ldk >= K")
        return out

    def unpack_int4(self, packed: np.ndarray, K: int, N: int) -> np.ndarray:
        q = np.zeros((N, K), dtype=np.int8)
        rc = self._unpack(np.ascontiguousarray(packed).reshape(-1), K // 2, K, N, q.reshape(-1), K)
        if rc != 0:
            raise ValueError("unpack_int4: K must be a multiple of 8")
        return q

    def offset_u8(self, q: np.ndarray) -> np.ndarray:
        out = np.empty(q.size, dtype=np.uint8)
        self._offs(np.ascontiguousarray(q).ravel(), out, q.size)
        return out.reshape(q.shape)


def load_port() -> Checker:
    return Checker(build_port(), "wqo_", "port")


def load_ref() -> Checker | None:
    path = build_ref()
    if path is None or not os.path.exists(path):
        return None
    return Checker(path, "ckref_", "reference")
