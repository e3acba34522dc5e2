"""CPU-only checks of the C-ABI surface: the library loads, exports every declared symbol,
validates problems and packs int4 on the host.  No GPU call is made here."""
import ctypes as C
import os
import re

import numpy as np

from composable_kernel_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "wqgemm.h")).read()
    declared = set(re.findall(r"\b(wqgemm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert hasattr(capi.lib, name), name
    assert "gfx950" in capi.version()


def _problem(**kw):
    fake = 0x10000  # never dereferenced by validation
    d = dict(a=fake, b=fake, scale=fake, c=fake, M=4096, N=4096, K=4096)
    d.update(kw)
    return capi.fpAintB_problem(d.pop("a"), d.pop("b"), d.pop("scale"), d.pop("c"), d.pop("M"), d.pop("N"),
                                d.pop("K"), **d)


def test_is_supported_and_routes():
    p = _problem()
    assert capi.is_supported(p)
    assert capi.kernel_name(p) == "wq_gemm_mfma_256"      # 256 tiles: one 256 x 256 workgroup per CU
    assert "DeviceFpAintBGemm_Mfma_gfx950" in capi.type_string(p)
    assert capi.kernel_name(_problem(flags=capi.FLAG_TILE_128)) == "wq_gemm_mfma_128"
    assert capi.kernel_name(_problem(M=512, N=512)) == "wq_gemm_mfma_128"   # too few 256-tiles to fill the chip
    assert capi.kernel_name(_problem(M=512, N=512, flags=capi.FLAG_TILE_256)) == "wq_gemm_mfma_256"
    assert not capi.is_supported(_problem(M=512, N=384, flags=capi.FLAG_TILE_256))   # N must be whole 256-tiles
    # M need not be: the last row tile of the tile kernels may be partial; up to 64 rows go to the small-M kernel
    assert capi.kernel_name(_problem(M=384, N=512, flags=capi.FLAG_TILE_256)) == "wq_gemm_mfma_256"
    assert capi.kernel_name(_problem(M=4000)) == "wq_gemm_mfma_256"
    assert capi.kernel_name(_problem(M=100)) == "wq_gemm_mfma_128"
    p = _problem(M=100)
    assert capi.workspace_bytes(p) > 0 and "KBatch" in capi.type_string(p)   # short grid: planner-chosen split-K
    assert capi.kernel_name(_problem(M=33)) == "wq_gemm_skinny"
    assert capi.kernel_name(_problem(M=2)) == "wq_gemm_skinny_ws"
    assert capi.kernel_name(_problem(M=16)) == "wq_gemm_skinny_wsc"
    assert not capi.is_supported(_problem(flags=1 << 20))   # unknown flag bits are rejected
    # N need only be a multiple of 4 for the 128-tile kernel (partial last column tile)
    assert capi.kernel_name(_problem(M=300, N=16)) == "wq_gemm_mfma_128"
    assert capi.kernel_name(_problem(M=300, N=1000)) == "wq_gemm_mfma_128"
    # ragged shapes fall back to the generic HIP kernel (still supported)
    p = _problem(M=33, N=17, K=72, lda=80, ldb=88, ldc=24)
    assert capi.is_supported(p) and capi.kernel_name(p) == "wq_gemm_generic"
    # packed int4 needs K % 8 == 0 and a K-contiguous B
    assert not capi.is_supported(_problem(K=100, lda=100, ldb=104))
    assert not capi.is_supported(_problem(b_layout=capi.ROW_MAJOR))
    # null pointers / bad sizes are invalid, not crashes
    assert not capi.is_supported(_problem(a=None))
    assert not capi.is_supported(_problem(M=0))
    assert not capi.is_supported(_problem(lda=0))
    assert not capi.is_supported(_problem(ldc=100))
    # an input stride below the row length aliases rows (the reference's descriptors allow it): generic kernel
    assert capi.kernel_name(_problem(lda=100)) == "wq_gemm_generic"
    # C dtype must follow A for the 16-bit paths
    assert not capi.is_supported(_problem(c_dtype=capi.BF16))
    # launching an unsupported problem returns a status, never raises from C
    bad = _problem(K=100, lda=100, ldb=104)
    assert capi.lib.wqgemm_launch(C.byref(bad), None, None) == capi.ERR_UNSUPPORTED
    assert b"int4" in capi.lib.wqgemm_last_error()


def _i8_problem(**kw):
    fake = 0x10000
    d = dict(a=fake, b=fake, scale=None, c=fake, M=4096, N=4096, K=4096, a_dtype=capi.I8, b_dtype=capi.I8,
             c_dtype=capi.I8, epilogue=capi.EPI_ACT_MUL_CLAMP, requant=0.03)
    d.update(kw)
    return capi.fpAintB_problem(d.pop("a"), d.pop("b"), d.pop("scale"), d.pop("c"), d.pop("M"), d.pop("N"),
                                d.pop("K"), **d)


def test_int8_requant_routes():
    """int8 x int8 -> int8 behind the (Add_)Activation_Mul_Clamp epilogues: which kernel serves which shape, and
    what the validation rejects."""
    assert capi.kernel_name(_i8_problem()) == "wq_gemm_i8_mfma_256"
    assert capi.kernel_name(_i8_problem(M=256, N=256, K=256)) == "wq_gemm_i8_mfma_128"     # the example's size
    assert capi.kernel_name(_i8_problem(M=256, N=256, K=256, flags=capi.FLAG_TILE_256)) == "wq_gemm_i8_mfma_256"
    assert capi.kernel_name(_i8_problem(M=1000)) == "wq_gemm_i8_mfma_128"                  # partial last row tile
    assert capi.kernel_name(_i8_problem(M=16)) == "wq_gemm_i8_skinny"                      # decode: one pass over the weights
    assert capi.kernel_name(_i8_problem(M=16, flags=capi.FLAG_TILE_128)) == "wq_gemm_i8_mfma_128"
    assert capi.kernel_name(_i8_problem(M=17)) == "wq_gemm_i8_mfma_128"
    assert capi.workspace_bytes(_i8_problem(M=4, N=1024)) == 4 * 4 * 1024 * 4             # 64 workgroups: four K slices of 1024
    assert capi.kernel_name(_i8_problem(N=1000)) == "wq_gemm_i8_mfma_128"                  # partial last column tile
    assert capi.kernel_name(_i8_problem(N=1002)) == "wq_gemm_generic"
    assert capi.kernel_name(_i8_problem(K=200, lda=200, ldb=200)) == "wq_gemm_generic"
    assert capi.kernel_name(_i8_problem(b_layout=capi.ROW_MAJOR)) == "wq_gemm_generic"     # mk_kn
    assert capi.kernel_name(_i8_problem(a_layout=capi.COL_MAJOR)) == "wq_gemm_generic"     # km_nk
    assert capi.is_supported(_i8_problem(epilogue=capi.EPI_RELU_MUL_CLAMP))
    # the bias variants need their int32 bias vector
    assert not capi.is_supported(_i8_problem(epilogue=capi.EPI_ADD_RELU_MUL_CLAMP))
    assert b"bias" in capi.lib.wqgemm_last_error()
    assert capi.is_supported(_i8_problem(epilogue=capi.EPI_ADD_RELU_MUL_CLAMP, d0=0x10000))
    assert capi.is_supported(_i8_problem(epilogue=capi.EPI_ADD_ACT_MUL_CLAMP, d0=0x10000))
    # int8 A only pairs with int8 B, int8 C and a requantising epilogue; no split-K
    assert not capi.is_supported(_i8_problem(epilogue=capi.EPI_PASSTHROUGH))
    assert not capi.is_supported(_i8_problem(c_dtype=capi.F16))
    assert not capi.is_supported(_i8_problem(b_dtype=capi.I4_PACKED))
    assert not capi.is_supported(_i8_problem(k_batch=2))
    assert capi.workspace_bytes(_i8_problem()) == 0


def test_splitk_workspace_and_kread():
    assert capi.splitk_kread(65536, 8) == 8192
    assert capi.splitk_kread(437, 3, 8) == 152
    p = _problem(k_batch=4)
    assert capi.workspace_bytes(p) == 4 * 4096 * 4096 * 4
    assert capi.workspace_bytes(_problem()) == 0
    # a K-slice launch writes fp32 partials straight to C: no workspace, c_dtype must be F32
    p = _problem(k_begin=8192, k_end=16384, K=65536, c_dtype=capi.F32)
    assert capi.is_supported(p) and capi.workspace_bytes(p) == 0
    assert not capi.is_supported(_problem(k_begin=8192, k_end=16384, K=65536))
    # split-K without a workspace is reported as such
    p = _problem(k_batch=2)
    assert capi.lib.wqgemm_launch(C.byref(p), None, None) == capi.ERR_WORKSPACE


def test_pack_int4_host_matches_checker(port):
    rng = np.random.default_rng(2)
    K, N = 128, 7
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    np.testing.assert_array_equal(capi.pack_int4_host(q, K, N), port.pack_int4(q, K, N))
    bad = q.copy()
    bad[3, 5] = -9
    try:
        capi.pack_int4_host(bad, K, N)
        raise AssertionError("out-of-range value accepted")
    except capi.WqError as e:
        assert e.status == capi.ERR_INVALID


def test_missing_library_fails_loudly(tmp_path):
    """There is no CPU fallback: without libwqgemm.so the binding must refuse to import (checked in a child process
    that is pointed at a path holding no library)."""
    import subprocess
    import sys
    env = dict(os.environ, WQGEMM_LIB=str(tmp_path / "libwqgemm.so"))
    r = subprocess.run([sys.executable, "-c", "import composable_kernel_amd.capi"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode != 0 and "ImportError" in r.stderr


def test_product_never_touches_the_checker():
    """The CPU checker under oracle/ is test infrastructure: nothing in the package, the C++ sources, the headers or
    the examples may import, include or link it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for top in ("composable_kernel_amd", "include", "examples"):
        for dirpath, _, files in os.walk(os.path.join(root, top)):
            if "_ref_build" in dirpath:
                continue
            for f in files:
                if not f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")) and f != "Makefile":
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"\boracle\b|wq_oracle|libck_ref", text):
                    offenders.append(os.path.relpath(os.path.join(dirpath, f), root))
    assert not offenders, offenders


def test_reduce_finalize_validates_epilogue_operands():
    """wqgemm_reduce_finalize is a public entry point of its own: it must refuse an epilogue whose D vectors are
    missing, and the int8 requantisation kinds (which have no split-K form), before anything is launched."""
    fake = 0x10000
    for epi in (capi.EPI_MULTIPLY_MULTIPLY, capi.EPI_ADD, capi.EPI_MULTIPLY_FASTGELU, capi.EPI_MULTIPLY_ADD_FASTGELU):
        p = _problem(M=64, N=64, K=64, epilogue=epi, d0=None, d1=None)
        assert capi.lib.wqgemm_reduce_finalize(C.byref(p), fake, 1, None) == capi.ERR_INVALID
        assert b"needs d0" in capi.lib.wqgemm_last_error()
    p = _problem(M=64, N=64, K=64, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=fake, d1=None)
    assert capi.lib.wqgemm_reduce_finalize(C.byref(p), fake, 1, None) == capi.ERR_INVALID
    for epi in (capi.EPI_ACT_MUL_CLAMP, capi.EPI_RELU_MUL_CLAMP, capi.EPI_ADD_ACT_MUL_CLAMP, 99):
        p = _problem(M=64, N=64, K=64, epilogue=epi, d0=fake)
        assert capi.lib.wqgemm_reduce_finalize(C.byref(p), fake, 1, None) == capi.ERR_UNSUPPORTED
    p = _problem(M=64, N=64, K=64, epilogue=capi.EPI_ADD, d0=fake, ldd0=-1)
    assert capi.lib.wqgemm_reduce_finalize(C.byref(p), fake, 1, None) == capi.ERR_INVALID


def test_fp8_activations_refuse_a_weight_scale():
    """The fp8 kernels take their scales through d0 / d1; a weight scale would be dropped, so it is refused."""
    fake = 0x10000
    kw = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=fake, d1=fake)
    assert capi.is_supported(_problem(scale=None, **kw))
    assert not capi.is_supported(_problem(scale=fake, **kw))
    assert b"scale must be NULL" in capi.lib.wqgemm_last_error()
    assert not capi.is_supported(_problem(scale=fake, flags=capi.FLAG_GENERIC, **kw))


def test_row_major_weights_with_split_k_route_and_workspace():
    """Row-major (K, N) int8 weights with the caller's k_batch (example/35's bf16A_i8B problem): matrix-core kernel
    over a K-contiguous copy that follows the fp32 slabs in the workspace."""
    M, N, K = 256, 1024, 512
    kw = dict(M=M, N=N, K=K, a_dtype=capi.BF16, b_dtype=capi.I8, b_layout=capi.ROW_MAJOR, scale=None)
    for kb in (2, 3, 5, 8):
        p = _problem(k_batch=kb, **kw)
        assert capi.is_supported(p)
        assert capi.kernel_name(p) == "wq_transpose_b8+wq_gemm_mfma_128"
        assert capi.workspace_bytes(p) == kb * M * N * 4 + K * N
        assert capi.lib.wqgemm_launch(C.byref(p), None, None) == capi.ERR_WORKSPACE
    p = _problem(k_batch=1, **kw)
    assert capi.workspace_bytes(p) == K * N
    # a K that is not whole 64-deep tiles stays on the general kernel, slabs only
    p = _problem(k_batch=2, **dict(kw, K=437, lda=437))
    assert capi.kernel_name(p) == "wq_gemm_generic" and capi.workspace_bytes(p) == 2 * M * N * 4


def test_ck_factory_registrations_link():
    """The factory specialisations this engine adds next to CK's resolve to composable_kernel_amd/ck_instances/ (binary
    built only where the CK tree is mounted; it constructs the operators and touches no GPU)."""
    import subprocess

    import pytest
    exe = os.path.join(ROOT, "examples", "_ref_build", "ck_factory_check")
    if not os.path.exists(exe):
        pytest.skip("ck_factory_check not built (needs the CK tree)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "found 8 instances" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("DeviceGemmSplitK_WeightQuant_gfx950") == 4


def test_preshuffle_int4_host_matches_checker(port):
    """The pre-shuffled weight order: the library's host re-layout against the checker's element-wise statement of the
    format, with and without a padded K pitch; bad sizes refused."""
    rng = np.random.default_rng(3)
    for K, N in ((64, 32), (256, 96), (128, 256)):
        q = rng.integers(-8, 8, (N, K)).astype(np.int8)
        packed = port.pack_int4(q, K, N)
        np.testing.assert_array_equal(capi.preshuffle_int4_host(packed, K, N), port.preshuffle_int4(packed, K, N))
    K, N, ldk = 128, 64, 192
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    packed = port.pack_int4(q, K, N)
    out = np.zeros(N * ldk // 2, np.uint8)
    capi.check(capi.lib.wqgemm_preshuffle_int4_host(packed.ctypes.data, K // 2, K, N, out.ctypes.data, ldk))
    np.testing.assert_array_equal(out, port.preshuffle_int4(packed, K, N, ldk))
    for bad_k, bad_n, bad_ldk in ((96, 32, 96), (64, 48, 64), (64, 32, 32), (64, 32, 96)):
        assert capi.lib.wqgemm_preshuffle_int4_host(packed.ctypes.data, bad_k // 2, bad_k, bad_n, out.ctypes.data,
                                                    bad_ldk) == capi.ERR_INVALID


def test_preshuffle_int8_host_matches_checker(port):
    rng = np.random.default_rng(4)
    for K, N in ((64, 16), (256, 96), (128, 256)):
        b = rng.integers(-128, 128, (N, K)).astype(np.int8)
        np.testing.assert_array_equal(capi.preshuffle_int8_host(b, K, N), port.preshuffle_int8(b, K, N))
    K, N, ldk = 128, 64, 192
    b = rng.integers(-128, 128, (N, K)).astype(np.int8)
    out = np.zeros(N * ldk, np.int8)
    capi.check(capi.lib.wqgemm_preshuffle_int8_host(b.ctypes.data, K, K, N, out.ctypes.data, ldk))
    np.testing.assert_array_equal(out, port.preshuffle_int8(b, K, N, ldk))
    for bad_k, bad_n, bad_ldk in ((96, 32, 96), (64, 24, 64), (64, 32, 32), (64, 32, 96)):
        assert capi.lib.wqgemm_preshuffle_int8_host(b.ctypes.data, bad_k, bad_k, bad_n, out.ctypes.data, bad_ldk) == capi.ERR_INVALID


def test_preshuffled_weights_routes():
    # read by every matrix-core kernel family; nothing falls back silently onto the general kernel
    pre = dict(b_dtype=capi.I4_PRESHUFFLED)
    assert capi.kernel_name(_problem(**pre)) == "wq_gemm_mfma_256"
    assert "I4_PRESHUFFLED" in capi.type_string(_problem(**pre))
    assert capi.kernel_name(_problem(M=8, **pre)).startswith("wq_gemm_skinny_ws")
    assert capi.kernel_name(_problem(M=48, **pre)) == "wq_gemm_skinny"
    assert capi.kernel_name(_problem(M=512, N=512, **pre)) == "wq_gemm_mfma_128"     # a short grid, as for the plain order
    assert capi.kernel_name(_problem(N=4096 + 128, **pre)) == "wq_gemm_mfma_128"     # N % 256 != 0
    assert capi.kernel_name(_problem(M=100, N=4096 + 32, **pre)) == "wq_gemm_mfma_128"
    assert capi.kernel_name(_problem(flags=capi.FLAG_TILE_128, **pre)) == "wq_gemm_mfma_128"
    assert capi.kernel_name(_problem(a_dtype=capi.BF16, **pre)) == "wq_gemm_mfma_256"
    assert capi.kernel_name(_problem(scale_group_k=128, **pre)) == "wq_gemm_mfma_256"
    assert capi.kernel_name(_problem(flags=capi.FLAG_GENERIC, **pre)) == "wq_gemm_generic"
    assert not capi.is_supported(_problem(N=4096 + 16, **pre))                  # N % 32 != 0: not even a layout
    assert not capi.is_supported(_problem(K=4096 + 32, ldb=4096 + 64, **pre))   # K % 64 != 0
    assert not capi.is_supported(_problem(ldb=4096 + 32, **pre))                # K pitch must be whole 64-k blocks
    assert capi.is_supported(_problem(ldb=4096 + 64, **pre))
    assert not capi.is_supported(_problem(scale_group_k=32, **pre))             # groups the tile kernels cannot take
    assert b"matrix-core" in capi.lib.wqgemm_last_error()
    # fp8 activations: the 256-tile kernel or nothing
    f8 = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, scale=None, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=0x10000, d1=0x10000)
    assert capi.kernel_name(_problem(**f8, **pre)) == "wq_gemm_f8_mfma_256"
    assert not capi.is_supported(_problem(N=4096 + 128, **f8, **pre))
    # 8-bit weights in their pre-shuffled order: the same rules; not an operand of the int8 x int8 requantised GEMM
    pre8 = dict(b_dtype=capi.I8_PRESHUFFLED)
    assert capi.kernel_name(_problem(a_dtype=capi.BF16, N=11008, **pre8)) == "wq_gemm_mfma_256"
    assert "I8_PRESHUFFLED" in capi.type_string(_problem(**pre8))
    assert capi.kernel_name(_problem(M=8, **pre8)).startswith("wq_gemm_skinny_ws")
    assert capi.kernel_name(_problem(N=4096 + 128, **pre8)) == "wq_gemm_mfma_128"
    assert capi.kernel_name(_problem(flags=capi.FLAG_GENERIC, **pre8)) == "wq_gemm_generic"
    assert not capi.is_supported(_problem(b_layout=capi.ROW_MAJOR, **pre8))
    assert not capi.is_supported(_problem(a_dtype=capi.I8, c_dtype=capi.I8, scale=None, epilogue=capi.EPI_ACT_MUL_CLAMP,
                                          requant=0.03, **pre8))


def test_group_wise_scale_validation_and_routes():
    # groups of a multiple of 64 k: the matrix-core kernels; anything else: the general kernel
    assert capi.kernel_name(_problem(scale_group_k=128)) == "wq_gemm_mfma_256"
    assert "ScaleGroupK128" in capi.type_string(_problem(scale_group_k=128))
    assert capi.kernel_name(_problem(M=512, N=512, scale_group_k=64)) == "wq_gemm_mfma_128"
    assert capi.kernel_name(_problem(scale_group_k=32)) == "wq_gemm_generic"
    # small M: the streaming kernels take groups of a power of two that cover a lane's 32 (int4) or 16 (int8) k per super-step
    assert capi.kernel_name(_problem(M=8, scale_group_k=128)).startswith("wq_gemm_skinny_ws")
    assert capi.kernel_name(_problem(M=48, scale_group_k=32)) == "wq_gemm_skinny"
    assert capi.kernel_name(_problem(M=8, scale_group_k=192)) == "wq_gemm_mfma_128"      # not a power of two
    assert capi.kernel_name(_problem(M=8, scale_group_k=16)) == "wq_gemm_generic"        # inside a lane's 32 k
    assert capi.kernel_name(_problem(M=8, scale_group_k=16, b_dtype=capi.I8)).startswith("wq_gemm_skinny")
    # the 256-tile kernel moves scale rows four bytes at a time: odd row strides stay on the 128-tile kernel
    assert capi.kernel_name(_problem(scale_group_k=128, ld_scale=4097)) == "wq_gemm_mfma_128"
    assert capi.kernel_name(_problem(scale_group_k=128, ld_scale=4104)) == "wq_gemm_mfma_256"
    assert not capi.is_supported(_problem(scale_group_k=128, ld_scale=4000))            # shorter than a row of N scales
    assert not capi.is_supported(_problem(scale_group_k=-1))
    assert not capi.is_supported(_problem(scale_group_k=128, scale=None))
    assert capi.kernel_name(_problem(scale_group_k=128, a_dtype=capi.BF16)) == "wq_gemm_mfma_256"
    assert capi.kernel_name(_problem(M=8, scale_group_k=128, a_dtype=capi.BF16, b_dtype=capi.I8)).startswith("wq_gemm_skinny")
    f8 = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=0x10000, d1=0x10000)
    assert not capi.is_supported(_problem(scale_group_k=128, **f8))                     # fp8: scales live in the epilogue


def test_problem_struct_size_handshake():
    # the binding refuses a library built with another wqgemm_problem (it checks at load time); here: they agree
    assert capi.lib.wqgemm_problem_size() == C.sizeof(capi.Problem)


def test_operands_beyond_2_gib_stay_off_the_32_bit_kernels():
    """The fp8 kernel and the 32x32-MFMA kernel address A and B with 32-bit byte offsets, so an operand above 2 GiB
    (the bound of include/ck/tensor_operation/gpu/grid/gridwise_fpAintB_gemm_wmma.hpp:501-507) is kept away from
    them; the kernels with 64-bit addressing take such operands."""
    f8 = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, scale=None, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=0x10000, d1=0x10000)
    assert capi.kernel_name(_problem(M=(1 << 19) - 256, **f8)) == "wq_gemm_f8_mfma_256"  # A just below 2 GiB
    assert capi.kernel_name(_problem(M=1 << 19, **f8)) == "wq_gemm_generic"              # A of exactly 2 GiB
    assert capi.kernel_name(_problem(N=1 << 20, K=8192, lda=8192, ldb=8192, **f8)) == "wq_gemm_generic"   # B = 4 GiB
    assert capi.kernel_name(_problem(N=1 << 20, K=8192, lda=8192, ldb=8192, b_dtype=capi.I4_PRESHUFFLED)) == "wq_gemm_mfma_256"
    assert capi.kernel_name(_problem(M=1 << 20)) == "wq_gemm_mfma_256"                   # fp16 A of 8 GiB: 64-bit addressing
    assert capi.is_supported(_problem(M=(1 << 18) - 256, flags=capi.FLAG_MFMA_32X32))
    assert not capi.is_supported(_problem(M=1 << 20, flags=capi.FLAG_MFMA_32X32))
    assert b"2 GiB" in capi.lib.wqgemm_last_error()


def test_general_kernel_row_limit():
    # its grid holds M / 16 in a 16-bit dimension: refused by validation, not by a failed launch
    assert capi.is_supported(_problem(M=1048560, flags=capi.FLAG_GENERIC))
    assert not capi.is_supported(_problem(M=1048561, flags=capi.FLAG_GENERIC))
    assert b"1048560 rows" in capi.lib.wqgemm_last_error()
    assert capi.is_supported(_problem(M=1 << 21))       # the tile kernels index tiles in one 32-bit grid dimension


def test_caller_split_k_of_fp8_activations_stays_on_the_fp8_kernel():
    # slices are cut in the fp8 kernel's 128-deep K tiles, whatever k_batch is (2048 / 3 in 64-deep tiles would not be)
    f8 = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, scale=None, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=0x10000, d1=0x10000)
    for K in (1024, 2048, 4096):
        for kb in (2, 3, 5, 7):
            for b_dtype in (capi.I4_PACKED, capi.I4_PRESHUFFLED):
                p = _problem(K=K, lda=K, ldb=K, k_batch=kb, b_dtype=b_dtype, **f8)
                assert capi.kernel_name(p) == "wq_gemm_f8_mfma_256", (K, kb)


def test_planner_split_k_puts_mid_sized_m_on_the_256_tile_kernel():
    """A few hundred rows against a few thousand columns: too few 256-tiles for one per CU, so the planner splits K
    (optional workspace) up to about one workgroup per CU instead of falling back to 128-tiles."""
    p = _problem(M=768)                                    # 3 x 16 = 48 tiles -> 5 slices
    assert capi.kernel_name(p) == "wq_gemm_mfma_256" and "KBatch5" in capi.type_string(p)
    assert capi.workspace_bytes(p) == 5 * 768 * 4096 * 4
    p = _problem(M=1536)                                   # 96 tiles -> 2 slices
    assert capi.kernel_name(p) == "wq_gemm_mfma_256" and "KBatch2" in capi.type_string(p)
    p = _problem(M=512)                                    # 32 tiles at K = 4096: the 128-tile kernel, two workgroups per CU
    assert capi.kernel_name(p) == "wq_gemm_mfma_128" and "KBatch4" in capi.type_string(p)
    p = _problem(M=512, K=8192, lda=8192, ldb=8192)        # ... but the 256-tile kernel once K is long
    assert capi.kernel_name(p) == "wq_gemm_mfma_256" and "KBatch8" in capi.type_string(p)
    p = _problem(M=128, N=11008)                           # at most 128 rows stay on 128-tiles
    assert capi.kernel_name(p) == "wq_gemm_mfma_128"
    p = _problem(M=768, K=1024, lda=1024, ldb=1024)        # slices would be shorter than 768 k: no split, 128-tiles
    assert capi.kernel_name(p) == "wq_gemm_mfma_128"
    p = _problem(M=768, k_batch=2)                         # the caller's own split is left as it is
    assert capi.kernel_name(p) == "wq_gemm_mfma_128" and "KBatch2" in capi.type_string(p)


def test_fp8_activations_short_grids_split_k_and_small_m():
    """fp8 activations: every M runs on the fp8 matrix-core kernel (no general-kernel fallback for decode-sized
    batches), and a grid shorter than the chip gets the planner's own K split (optional workspace)."""
    f8 = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, scale=None, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=0x10000, d1=0x10000)
    # decode: the streaming kernel (one pass over B); it widens the bytes itself while the fp16 image fits its LDS budget
    # (no workspace then), else a first kernel writes the image to the workspace
    p = _problem(M=1, **f8)
    assert capi.kernel_name(p) == "wq_gemm_skinny_ws" and capi.workspace_bytes(p) == 0
    p = _problem(M=16, **f8)
    assert capi.kernel_name(p) == "wq_f8_fnuz_to_f16+wq_gemm_skinny_wsc" and capi.workspace_bytes(p) == (16 * 4096 + 4096) * 2
    assert capi.kernel_name(_problem(M=16, flags=capi.FLAG_PINGPONG, **f8)) == "wq_gemm_f8_mfma_256"   # a kernel asked for by name
    for M in (17, 64, 65, 512):
        p = _problem(M=M, **f8)
        assert capi.kernel_name(p) == "wq_gemm_f8_mfma_256"
        assert "KBatch" in capi.type_string(p) and "KBatch1>" not in capi.type_string(p), capi.type_string(p)
        assert capi.workspace_bytes(p) > 0
    for M in (2048, 4096):                                # half the chip and more: one slice
        p = _problem(M=M, **f8)
        assert "KBatch1>" in capi.type_string(p) and capi.workspace_bytes(p) == 0
