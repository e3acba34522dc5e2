"""Minimal ctypes view of RCCL's C API: what a C++ caller of ``wqgemm_splitk_multi_step`` holds -- a unique id and an
``ncclComm_t`` -- made from Python for the benchmark and the tests.

The library bound is the copy PyTorch ships (and has already loaded), so the process holds one RCCL and
``libwqgemm.so`` finds the same one at run time.  Nothing here moves data: the collectives are issued inside
``wqgemm_splitk_multi_step``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads its librccl first)

UNIQUE_ID_BYTES = 128  # NCCL_UNIQUE_ID_BYTES


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


def _load() -> C.CDLL:
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path if os.path.exists(path) else "librccl.so.1")
    lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    return lib


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RcclError(f"{what} failed: {lib().ncclGetErrorString(rc).decode()}")


def get_unique_id() -> bytes:
    """ncclGetUniqueId: 128 opaque bytes that rank 0 hands to every other rank (any transport)."""
    uid = UniqueId()
    _check(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return bytes(C.string_at(C.addressof(uid), UNIQUE_ID_BYTES))


def comm_init_rank(nranks: int, uid: bytes, rank: int) -> int:
    """ncclCommInitRank on the current device; returns the ncclComm_t as an integer.  Collective over the ranks."""
    assert len(uid) == UNIQUE_ID_BYTES
    u = UniqueId()
    C.memmove(C.addressof(u), uid, UNIQUE_ID_BYTES)
    comm = C.c_void_p()
    _check(lib().ncclCommInitRank(C.byref(comm), nranks, u, rank), "ncclCommInitRank")
    return int(comm.value)


def comm_destroy(comm: int) -> None:
    _check(lib().ncclCommDestroy(C.c_void_p(comm)), "ncclCommDestroy")
