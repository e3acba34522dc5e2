"""ctypes binding of the C ABI in ``include/wqgemm.h``.

Every function here forwards to ``libwqgemm.so``; nothing is computed in Python.  The library is
looked up in-tree (``composable_kernel_amd/csrc/libwqgemm.so``) and a missing library is an
ImportError -- there is deliberately no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own libamdhip64 (same soname as the system ROCm one).  Loading it first makes the
# dynamic loader bind libwqgemm.so to that already-loaded runtime, so the process holds exactly one
# HIP runtime and torch-allocated device pointers are valid in our launches.  Loading ours first
# would bring in a second runtime that cannot see the device.
import torch  # noqa: F401  (must precede the CDLL below)

HERE = os.path.dirname(os.path.abspath(__file__))
# WQGEMM_LIB names another build of the same library (A/B timing of two builds in one session)
LIB_PATH = os.environ.get("WQGEMM_LIB") or os.path.join(HERE, "csrc", "libwqgemm.so")

# wq_status
OK, ERR_UNSUPPORTED, ERR_INVALID, ERR_HIP, ERR_WORKSPACE = range(5)
# wq_dtype
F16, BF16, F8_FNUZ, F32, I8, U8_OFFSET, I4_PACKED, I32, F8_E4M3FN, I4_PRESHUFFLED, I8_PRESHUFFLED = range(11)
# wq_layout
ROW_MAJOR, COL_MAJOR = 0, 1
# wqgemm_problem.flags
FLAG_TILE_128, FLAG_TILE_256, FLAG_LOCKSTEP, FLAG_WAVES_2X4 = 1, 2, 4, 8
FLAG_DIRECT_STORE, FLAG_GROUP8, FLAG_DEQUANT_IN_L, FLAG_GENERIC, FLAG_F8_NO_SPECIALS = 32, 64, 512, 1024, 2048
FLAG_MFMA_32X32 = 4096
FLAG_PINGPONG = 8192
# wq_epilogue
(EPI_PASSTHROUGH, EPI_FASTGELU, EPI_MULTIPLY_ADD_FASTGELU, EPI_MULTIPLY_MULTIPLY,
 EPI_MULTIPLY_MULTIPLY_FASTGELU, EPI_ACT_MUL_CLAMP, EPI_ADD, EPI_ADD_FASTGELU, EPI_MULTIPLY,
 EPI_MULTIPLY_FASTGELU, EPI_RELU_MUL_CLAMP, EPI_ADD_ACT_MUL_CLAMP, EPI_ADD_RELU_MUL_CLAMP, EPI_MULTIPLY_ADD, EPI_CONVERT_BF16_RTN) = range(15)

# every symbol include/wqgemm.h declares
EXPORTS = (
    "wqgemm_version", "wqgemm_problem_size", "wqgemm_last_error", "wqgemm_is_supported", "wqgemm_workspace_bytes",
    "wqgemm_launch", "wqgemm_run_timed", "wqgemm_type_string", "wqgemm_kernel_name",
    "wqgemm_pack_int4", "wqgemm_pack_int4_host", "wqgemm_preshuffle_int4", "wqgemm_preshuffle_int4_host", "wqgemm_preshuffle_int8", "wqgemm_preshuffle_int8_host", "wqgemm_reduce_finalize", "wqgemm_splitk_kread",
    "wqgemm_run_timed_rotating", "wqgemm_launch_noalloc", "wqgemm_splitk_multi_workspace_bytes",
    "wqgemm_splitk_multi_step",
)


class Problem(C.Structure):
    """``struct wqgemm_problem``."""
    _fields_ = [
        ("a_dtype", C.c_int32), ("b_dtype", C.c_int32), ("scale_dtype", C.c_int32), ("c_dtype", C.c_int32),
        ("a_layout", C.c_int32), ("b_layout", C.c_int32), ("c_layout", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("a", C.c_void_p), ("b", C.c_void_p), ("scale", C.c_void_p), ("c", C.c_void_p),
        ("epilogue", C.c_int32), ("requant_scale", C.c_float),
        ("d0", C.c_void_p), ("d1", C.c_void_p),
        ("k_batch", C.c_int32), ("k_begin", C.c_int32), ("k_end", C.c_int32), ("flags", C.c_int32),
        ("ldd0", C.c_int32), ("ldd1", C.c_int32),
        ("scale_group_k", C.c_int32), ("ld_scale", C.c_int32),
        ("a_block_scale", C.c_void_p), ("b_block_scale", C.c_void_p),
        ("scale_block_m", C.c_int32), ("scale_block_n", C.c_int32), ("scale_block_k", C.c_int32),
        ("ld_a_scale", C.c_int32), ("ld_b_scale", C.c_int32),
    ]


class WqError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"wqgemm status {status}: {msg}")
        self.status = status


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C composable_kernel_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    P = C.POINTER(Problem)
    L.wqgemm_version.restype = C.c_char_p
    L.wqgemm_problem_size.restype = C.c_size_t
    if L.wqgemm_problem_size() != C.sizeof(Problem):
        raise ImportError(f"{LIB_PATH}: built with a wqgemm_problem of {L.wqgemm_problem_size()} bytes, this binding "
                          f"describes {C.sizeof(Problem)}: rebuild the library (make -C composable_kernel_amd/csrc)")
    L.wqgemm_last_error.restype = C.c_char_p
    L.wqgemm_is_supported.restype = C.c_int
    L.wqgemm_is_supported.argtypes = [P]
    L.wqgemm_workspace_bytes.restype = C.c_size_t
    L.wqgemm_workspace_bytes.argtypes = [P]
    L.wqgemm_launch.restype = C.c_int
    L.wqgemm_launch.argtypes = [P, C.c_void_p, C.c_void_p]
    L.wqgemm_launch_noalloc.restype = C.c_int
    L.wqgemm_launch_noalloc.argtypes = [P, C.c_void_p]
    L.wqgemm_run_timed.restype = C.c_int
    L.wqgemm_run_timed.argtypes = [P, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.wqgemm_run_timed_rotating.restype = C.c_int
    L.wqgemm_run_timed_rotating.argtypes = [P, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.wqgemm_splitk_multi_workspace_bytes.restype = C.c_size_t
    L.wqgemm_splitk_multi_workspace_bytes.argtypes = [P, C.c_int32]
    L.wqgemm_splitk_multi_step.restype = C.c_int
    L.wqgemm_splitk_multi_step.argtypes = [P, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.wqgemm_type_string.restype = C.c_int
    L.wqgemm_type_string.argtypes = [P, C.c_char_p, C.c_size_t]
    L.wqgemm_kernel_name.restype = C.c_int
    L.wqgemm_kernel_name.argtypes = [P, C.c_char_p, C.c_size_t]
    L.wqgemm_pack_int4.restype = C.c_int
    L.wqgemm_pack_int4.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.wqgemm_preshuffle_int4.restype = C.c_int
    L.wqgemm_preshuffle_int4.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.wqgemm_preshuffle_int4_host.restype = C.c_int
    L.wqgemm_preshuffle_int4_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    L.wqgemm_preshuffle_int8.restype = C.c_int
    L.wqgemm_preshuffle_int8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.wqgemm_preshuffle_int8_host.restype = C.c_int
    L.wqgemm_preshuffle_int8_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    L.wqgemm_pack_int4_host.restype = C.c_int
    L.wqgemm_pack_int4_host.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    L.wqgemm_reduce_finalize.restype = C.c_int
    L.wqgemm_reduce_finalize.argtypes = [P, C.c_void_p, C.c_int32, C.c_void_p]
    L.wqgemm_splitk_kread.restype = C.c_int32
    L.wqgemm_splitk_kread.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    return L


lib = _load()


def check(status: int) -> None:
    if status != OK:
        raise WqError(status, lib.wqgemm_last_error().decode())


def version() -> str:
    return lib.wqgemm_version().decode()


def is_supported(p: Problem) -> bool:
    return bool(lib.wqgemm_is_supported(C.byref(p)))


def workspace_bytes(p: Problem) -> int:
    return int(lib.wqgemm_workspace_bytes(C.byref(p)))


def launch(p: Problem, workspace: int | None = None, stream: int | None = None) -> None:
    check(lib.wqgemm_launch(C.byref(p), workspace, stream))


NOALLOC_SCRATCH_BYTES = 64 << 20


def launch_noalloc(p: Problem, stream: int | None = None) -> None:
    """wqgemm_launch with the library's built-in per-GPU scratch in place of a caller workspace."""
    check(lib.wqgemm_launch_noalloc(C.byref(p), stream))


def run_timed(p: Problem, workspace: int | None, stream: int | None, cold: int, repeat: int) -> float:
    ms = C.c_float(0.0)
    check(lib.wqgemm_run_timed(C.byref(p), workspace, stream, cold, repeat, C.byref(ms)))
    return float(ms.value)


def run_timed_rotating(p: Problem, workspace: int | None, stream: int | None, cold: int, repeat: int,
                       rotating_count: int) -> float:
    """Median ms over `repeat` individually timed launches cycling through `rotating_count` copies of A and B."""
    ms = C.c_float(0.0)
    check(lib.wqgemm_run_timed_rotating(C.byref(p), workspace, stream, cold, repeat, rotating_count, C.byref(ms)))
    return float(ms.value)


def splitk_multi_workspace_bytes(p: Problem, nranks: int) -> int:
    return int(lib.wqgemm_splitk_multi_workspace_bytes(C.byref(p), nranks))


def splitk_multi_step(p: Problem, comm: int | None, rank: int, nranks: int, workspace: int,
                      stream: int | None = None) -> None:
    """One rank's step of the multi-GPU split-K GEMM (comm: an ncclComm_t as an integer, or None for no exchange)."""
    check(lib.wqgemm_splitk_multi_step(C.byref(p), comm, rank, nranks, workspace, stream))


def type_string(p: Problem) -> str:
    buf = C.create_string_buffer(512)
    lib.wqgemm_type_string(C.byref(p), buf, 512)
    return buf.value.decode()


def kernel_name(p: Problem) -> str:
    buf = C.create_string_buffer(256)
    check(lib.wqgemm_kernel_name(C.byref(p), buf, 256))
    return buf.value.decode()


def pack_int4(q_ptr: int, ldq: int, K: int, N: int, out_ptr: int, ld_bytes: int, stream: int | None = None) -> None:
    check(lib.wqgemm_pack_int4(q_ptr, ldq, K, N, out_ptr, ld_bytes, stream))


def preshuffle_int4(packed_ptr: int, ld_bytes: int, K: int, N: int, out_ptr: int, ldk: int, stream: int | None = None) -> None:
    """WQ_I4_PACKED -> WQ_I4_PRESHUFFLED on the device (layout: include/wqgemm.h, wqgemm_preshuffle_int4)."""
    check(lib.wqgemm_preshuffle_int4(packed_ptr, ld_bytes, K, N, out_ptr, ldk, stream))


def preshuffle_int4_host(packed, K: int, N: int):
    """packed: numpy uint8 (N, K // 2) in WQ_I4_PACKED order -> flat numpy uint8 of N * K // 2 bytes in WQ_I4_PRESHUFFLED
    order (ldk = K)."""
    import numpy as np

    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    assert packed.shape == (N, K // 2)
    out = np.empty(N * K // 2, dtype=np.uint8)
    check(lib.wqgemm_preshuffle_int4_host(packed.ctypes.data, K // 2, K, N, out.ctypes.data, K))
    return out


def preshuffle_int8(b_ptr: int, ldb: int, K: int, N: int, out_ptr: int, ldk: int, stream: int | None = None) -> None:
    """WQ_I8 -> WQ_I8_PRESHUFFLED on the device (layout: include/wqgemm.h, wqgemm_preshuffle_int8)."""
    check(lib.wqgemm_preshuffle_int8(b_ptr, ldb, K, N, out_ptr, ldk, stream))


def preshuffle_int8_host(b, K: int, N: int):
    """b: numpy int8 (N, K), K contiguous -> flat numpy int8 of N * K bytes in WQ_I8_PRESHUFFLED order (ldk = K)."""
    import numpy as np

    b = np.ascontiguousarray(b, dtype=np.int8)
    assert b.shape == (N, K)
    out = np.empty(N * K, dtype=np.int8)
    check(lib.wqgemm_preshuffle_int8_host(b.ctypes.data, K, K, N, out.ctypes.data, K))
    return out


def pack_int4_host(q, K: int, N: int):
    """numpy int8 (N, K) [element (k,n) at n*K + k] -> numpy uint8 (N, K//2)."""
    import numpy as np
    q = np.ascontiguousarray(q, dtype=np.int8)
    out = np.zeros((N, K // 2), dtype=np.uint8)
    check(lib.wqgemm_pack_int4_host(q.ctypes.data, K, K, N, out.ctypes.data, K // 2))
    return out


def reduce_finalize(p: Problem, partials_ptr: int, k_batch: int, stream: int | None = None) -> None:
    check(lib.wqgemm_reduce_finalize(C.byref(p), partials_ptr, k_batch, stream))


def splitk_kread(K: int, k_batch: int, granule: int = 64) -> int:
    return int(lib.wqgemm_splitk_kread(K, k_batch, granule))


def fpAintB_problem(a, b, scale, c, M, N, K, *, a_dtype=F16, b_dtype=I4_PACKED, lda=None, ldb=None,
                    ldc=None, epilogue=EPI_PASSTHROUGH, d0=None, d1=None, k_batch=1, k_begin=0,
                    k_end=0, c_dtype=None, b_layout=COL_MAJOR, a_layout=ROW_MAJOR, requant=0.0, flags=0,
                    ldd0=0, ldd1=0, scale_group_k=0, ld_scale=0, a_block_scale=None, b_block_scale=None,
                    scale_block=(128, 128, 128), ld_a_scale=0, ld_b_scale=0) -> Problem:
    """Fill a Problem from raw device pointers (ints).  Strides default to packed tensors."""
    p = Problem()
    p.a_dtype, p.b_dtype, p.scale_dtype = a_dtype, b_dtype, a_dtype
    p.c_dtype = a_dtype if c_dtype is None else c_dtype
    p.a_layout, p.b_layout, p.c_layout = a_layout, b_layout, ROW_MAJOR
    p.M, p.N, p.K = M, N, K
    p.lda = lda if lda is not None else (K if a_layout == ROW_MAJOR else M)
    p.ldb = ldb if ldb is not None else (K if b_layout == COL_MAJOR else N)
    p.ldc = ldc if ldc is not None else N
    p.a, p.b, p.scale, p.c = a, b, scale, c
    p.epilogue, p.requant_scale = epilogue, requant
    p.d0, p.d1 = d0, d1
    p.k_batch, p.k_begin, p.k_end, p.flags = k_batch, k_begin, k_end, flags
    p.ldd0, p.ldd1 = ldd0, ldd1
    p.scale_group_k, p.ld_scale = scale_group_k, ld_scale
    p.a_block_scale, p.b_block_scale = a_block_scale, b_block_scale
    if a_block_scale is not None or b_block_scale is not None:
        p.scale_block_m, p.scale_block_n, p.scale_block_k = scale_block
        p.ld_a_scale, p.ld_b_scale = ld_a_scale, ld_b_scale
    return p
