"""GPU parity of what the split-K client operators and the full-matrix D operands rest on, through the C ABI:
row-major int8 weights with the caller's k_batch (example/35's bf16A_i8B problem), D tensors with a leading
dimension (client_example/30's bias form), and the rotating-buffer timing mode."""
import numpy as np
import pytest
import torch

from composable_kernel_amd import capi
from gpu_util import bf16_bits, bits_to_f32, check_err, dev, f16_bits, host_u16, run_fpaintb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kb", [1, 2, 3, 5, 8])
def test_example35_problem_golden(golden_gemm, kb):
    """bf16 A x row-major int8 B, no scale, KBatch from the reference's split-K test set
    (test/gemm_split_k/test_gemm_splitk_util.hpp:36-87): integer inputs bit-exact against the reference's output for
    every KBatch (all partial sums exact in fp32); decimal inputs within one bf16 ulp (fp32 summation order differs,
    the store truncates)."""
    g = golden_gemm
    M, N, K = 64, 128, 512
    kern = "wq_transpose_b8+wq_gemm_mfma_128"
    got = run_fpaintb(g["sk_a_int"], g["sk_brow"], None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                      b_layout=capi.ROW_MAJOR, ldb=N, k_batch=kb, flags=capi.FLAG_TILE_128, expect_kernel=kern)
    np.testing.assert_array_equal(got, g["sk_c_int"].reshape(M, N))
    got = run_fpaintb(g["sk_a_dec"], g["sk_brow_full"], None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                      b_layout=capi.ROW_MAJOR, ldb=N, k_batch=kb, flags=capi.FLAG_TILE_128, expect_kernel=kern)
    exp = g["sk_c_dec"].reshape(M, N)
    ulp = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    tiny = np.abs(bits_to_f32(got, capi.BF16) - bits_to_f32(exp, capi.BF16)) <= 1e-3
    assert (ulp[~tiny] <= 1).all(), ulp[~tiny].max()
    # the general kernel (ascending k inside a slice) agrees with the matrix-core route bit for bit on integers
    gen = run_fpaintb(g["sk_a_int"], g["sk_brow"], None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                      b_layout=capi.ROW_MAJOR, ldb=N, k_batch=kb, flags=capi.FLAG_GENERIC, expect_kernel="wq_gemm_generic")
    np.testing.assert_array_equal(gen, g["sk_c_int"].reshape(M, N))


@pytest.mark.parametrize("a_dtype", [capi.F16, capi.BF16])
def test_row_major_b_split_k_equals_k_contiguous(port, a_dtype):
    """Larger case on the 256-tile kernel: the transposed-in-workspace route with k_batch is bit-identical to the
    same call on K-contiguous weights, and matches the checker on integer inputs."""
    rng = np.random.default_rng(350 + a_dtype)
    M, N, K = 512, 512, 1024
    cvt = f16_bits if a_dtype == capi.F16 else bf16_bits
    q = rng.integers(-3, 4, (N, K)).astype(np.int8)
    brow = np.ascontiguousarray(q.T)
    a = cvt(rng.integers(-2, 3, (M, K)).astype(np.float32))
    for kb, flags, kern in ((4, capi.FLAG_TILE_256, "wq_gemm_mfma_256"), (3, 0, None)):
        col = run_fpaintb(a, q, None, M, N, K, a_dtype=a_dtype, b_dtype=capi.I8, k_batch=kb, flags=flags, expect_kernel=kern)
        row = run_fpaintb(a, brow, None, M, N, K, a_dtype=a_dtype, b_dtype=capi.I8, b_layout=capi.ROW_MAJOR, ldb=N,
                          k_batch=kb, flags=flags, expect_kernel=("wq_transpose_b8+" + kern) if kern else None)
        np.testing.assert_array_equal(row, col)
    if a_dtype == capi.BF16:
        exp = port.gemm_bf16_i8(a, q, None, M, N, K, K, 1, K, N, 0).reshape(M, N)
        np.testing.assert_array_equal(row, exp)


def test_full_matrix_d_golden(golden_gemm):
    """D handed over as a full (M, N) matrix: general kernel bit-exact against the reference's outputs for the
    non-FastGelu functors, one bf16 ulp for the FastGelu ones (device exp2 / rcp against host exp and division)."""
    g = golden_gemm
    M, N, K = 48, 64, 128
    kinds = {3: capi.EPI_ADD, 4: capi.EPI_ADD_FASTGELU, 5: capi.EPI_MULTIPLY, 6: capi.EPI_MULTIPLY_FASTGELU}

    def same(got, exp, gelu):
        exp = exp.reshape(M, N)
        if not gelu:
            np.testing.assert_array_equal(got, exp)
            return
        d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
        tiny = np.abs(bits_to_f32(got, capi.BF16) - bits_to_f32(exp, capi.BF16)) <= 1e-6
        assert (d[~tiny] <= 1).all()

    for ep, kind in kinds.items():
        got = run_fpaintb(g["bf16_a"], g["bf16_bcol"], g["bf16_s"], M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                          epilogue=kind, d0=g["bf16_dm0"], ldd0=N, flags=capi.FLAG_GENERIC, expect_kernel="wq_gemm_generic")
        same(got, g[f"bf16_dmat_e_col_ep{ep}"], ep in (4, 6))
    got = run_fpaintb(g["bf16_a"], g["bf16_bcol"], None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                      epilogue=capi.EPI_MULTIPLY_ADD_FASTGELU, d0=g["bf16_dm0"], ldd0=N, d1=g["bf16_dm1"], ldd1=N + 8,
                      flags=capi.FLAG_GENERIC, expect_kernel="wq_gemm_generic")
    same(got, g["bf16_dmat_e_col_ep2"], True)


def test_full_matrix_d_on_matrix_core_kernels(port):
    """Same operands on the 128- and 256-tile kernels (and through split-K's reduce kernel) against the checker."""
    rng = np.random.default_rng(77)
    M, N, K = 256, 256, 256
    q = rng.integers(-128, 128, (N, K)).astype(np.int8)
    a = bf16_bits(rng.uniform(-1, 1, (M, K)))
    s = bf16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N))
    dm = bf16_bits(rng.uniform(-0.5, 0.5, (M, N + 16)))
    for flags, kb in ((capi.FLAG_TILE_128, 1), (capi.FLAG_TILE_256, 1), (capi.FLAG_TILE_128, 2)):
        for ep, kind in ((3, capi.EPI_ADD), (6, capi.EPI_MULTIPLY_FASTGELU)):
            got = run_fpaintb(a, q, s, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, epilogue=kind, d0=dm, ldd0=N + 16,
                              flags=flags, k_batch=kb)
            exp = port.gemm_bf16_i8_dmat(a, q, s, M, N, K, K, 1, K, N, ep, dm, N + 16).reshape(M, N)
            check_err(bits_to_f32(got, capi.BF16), bits_to_f32(exp, capi.BF16), 1e-2, 1e-3)
    # MultiplyMultiply with full matrices: d0 row-major (ld N), d1 column-major (ld M), fp8 E4M3FN activations
    a8 = rng.choice(np.array([0x00, 0x38, 0x40, 0xB8, 0xC0], np.uint8), (M, K))   # E4M3FN 0, 1, 2, -1, -2
    q4 = rng.integers(-8, 8, (N, K)).astype(np.int8)
    from gpu_util import pack_on_device
    d0m = rng.uniform(0.5, 1.0, (M, N)).astype(np.float32)
    d1m = rng.uniform(0.5, 1.0, (N, M)).astype(np.float32)   # column-major (M, N): element (m, n) at m + n*M
    got = run_fpaintb(a8, pack_on_device(q4, K, N), None, M, N, K, a_dtype=capi.F8_E4M3FN, epilogue=capi.EPI_MULTIPLY_MULTIPLY,
                      d0=d0m, ldd0=N, d1=d1m, ldd1=M, expect_kernel="wq_gemm_f8_mfma_256")
    ones = np.ones(max(M, N), np.float32)
    base = run_fpaintb(a8, pack_on_device(q4, K, N), None, M, N, K, a_dtype=capi.F8_E4M3FN,
                       epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=ones[:N], d1=ones[:M], expect_kernel="wq_gemm_f8_mfma_256")
    # integer accumulators: the unit-scale run returns them exactly where |acc| < 2^11; there the full-matrix result
    # must be half((acc * d0[m, n]) * d1[m, n]) to the bit
    acc = bits_to_f32(base, capi.F16)
    ok = np.abs(acc) < 2048
    assert ok.mean() > 0.9
    exp = ((acc * d0m) * d1m.T).astype(np.float16)
    np.testing.assert_array_equal(got[ok], exp.view(np.uint16)[ok])


def test_rotating_timed_run_matches_plain():
    """wqgemm_run_timed_rotating (StreamConfig::flush_cache's mode): results unchanged, time within a factor of the
    warm-cache mean (it cannot be faster than ~0.8x of it, nor wildly slower for a 1024^3 problem)."""
    rng = np.random.default_rng(5)
    M = N = K = 1024
    a_t = dev(f16_bits(rng.uniform(-1, 1, (M, K))))
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    from gpu_util import pack_on_device
    b_t = pack_on_device(q, K, N)
    s_t = dev(f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N)))
    c_t = torch.zeros((M, N), dtype=torch.int16, device="cuda")
    p = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), s_t.data_ptr(), c_t.data_ptr(), M, N, K)
    ws = torch.empty(max(capi.workspace_bytes(p), 1), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    capi.launch(p, ws.data_ptr(), st)
    torch.cuda.synchronize()
    ref = host_u16(c_t).copy()
    plain = capi.run_timed(p, ws.data_ptr(), st, 5, 20)
    rot = capi.run_timed_rotating(p, ws.data_ptr(), st, 5, 21, 4)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host_u16(c_t), ref)
    assert 0.5 * plain < rot < 20 * plain + 0.05, (plain, rot)
    assert p.a == a_t.data_ptr() and p.b == b_t.data_ptr()     # the caller's problem is not modified
    with pytest.raises(capi.WqError):
        capi.run_timed_rotating(p, ws.data_ptr(), st, 1, 3, 65)


def test_splitk_multi_step_c_abi(port):
    """wqgemm_splitk_multi_step, the C-ABI form of the multi-GPU flow: (a) through a real one-rank RCCL communicator
    (reduce-scatter and all-gather are issued) the result is the single-pass kernel's, bit for bit; (b) without
    exchange, as ranks 0 and 1 of two, each row block lands at its row offset with its per-row epilogue operands."""
    rng = np.random.default_rng(91)
    M, N, K = 512, 512, 1024
    from gpu_util import pack_on_device
    a_t = dev(f16_bits(rng.uniform(-1, 1, (M, K))))
    q = rng.integers(-8, 8, (N, K)).astype(np.int8)
    b_t = pack_on_device(q, K, N)
    s_t = dev(f16_bits(rng.uniform(2.0 ** -8, 2.0 ** -4, N)))
    st = torch.cuda.current_stream().cuda_stream
    c_ref = torch.zeros((M, N), dtype=torch.int16, device="cuda")
    p_ref = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), s_t.data_ptr(), c_ref.data_ptr(), M, N, K,
                                 flags=capi.FLAG_TILE_256)
    capi.launch(p_ref, None, st)
    torch.cuda.synchronize()

    from composable_kernel_amd import rccl
    comm = rccl.comm_init_rank(1, rccl.get_unique_id(), 0)
    c = torch.zeros((M, N), dtype=torch.int16, device="cuda")
    p = capi.fpAintB_problem(a_t.data_ptr(), b_t.data_ptr(), s_t.data_ptr(), c.data_ptr(), M, N, K, flags=capi.FLAG_TILE_256)
    ws = torch.empty(capi.splitk_multi_workspace_bytes(p, 1), dtype=torch.uint8, device="cuda")
    assert ws.numel() == 2 * M * N * 4
    capi.splitk_multi_step(p, comm, 0, 1, ws.data_ptr(), st)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host_u16(c), host_u16(c_ref))
    rccl.comm_destroy(comm)

    # (b) two ranks rehearsed in one process, MultiplyMultiply with per-row d1: fp8 activations
    a8 = dev(rng.choice(np.array([0x00, 0x38, 0x40, 0xB8, 0xC0], np.uint8), (M, K)))
    d0 = dev(rng.uniform(0.5, 1.0, N).astype(np.float32))
    d1 = dev(rng.uniform(0.5, 1.0, M).astype(np.float32))
    kw = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0.data_ptr(), d1=d1.data_ptr())
    c_ref.zero_()
    capi.launch(capi.fpAintB_problem(a8.data_ptr(), b_t.data_ptr(), None, c_ref.data_ptr(), M, N, K, **kw), None, st)
    c.fill_(0x7BFF)
    p2 = capi.fpAintB_problem(a8.data_ptr(), b_t.data_ptr(), None, c.data_ptr(), M, N, K, **kw)
    ws2 = torch.empty(capi.splitk_multi_workspace_bytes(p2, 2), dtype=torch.uint8, device="cuda")
    capi.splitk_multi_step(p2, None, 1, 2, ws2.data_ptr(), st)
    torch.cuda.synchronize()
    assert (host_u16(c)[: M // 2] == 0x7BFF).all()          # rank 1 wrote only its rows
    capi.splitk_multi_step(p2, None, 0, 2, ws2.data_ptr(), st)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host_u16(c), host_u16(c_ref))
    # refusals: padded C, M not divisible
    bad = capi.fpAintB_problem(a8.data_ptr(), b_t.data_ptr(), None, c.data_ptr(), M, N, K, ldc=N + 8, **kw)
    with pytest.raises(capi.WqError):
        capi.splitk_multi_step(bad, None, 0, 2, ws2.data_ptr(), st)
    with pytest.raises(capi.WqError):
        capi.splitk_multi_step(p2, None, 0, 3, ws2.data_ptr(), st)


@pytest.mark.parametrize("tag", ["k320", "k437"])
@pytest.mark.parametrize("kb", [1, 2, 3, 5, 8])
def test_reference_splitk_shapes_golden(tag, kb):
    """The K extents of the reference's split-K test that are not whole 64-k tiles per slice (K = 320: five tiles; PaddK
    K = 437: not a multiple of anything), KBatch from its list, against outputs of the reference's own ReferenceGemm:
    bit-exact on the integer set, within one bf16 step on the decimal set.  Whatever kernel the planner picks."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "splitk_shapes.npz"))
    M, N, K = (int(v) for v in g[f"{tag}_shape"])
    got = run_fpaintb(g[f"{tag}_a_int"], g[f"{tag}_brow"], None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                      b_layout=capi.ROW_MAJOR, ldb=N, k_batch=kb)
    np.testing.assert_array_equal(got, g[f"{tag}_c_int"].reshape(M, N))
    got = run_fpaintb(g[f"{tag}_a_dec"], g[f"{tag}_brow_full"], None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8,
                      b_layout=capi.ROW_MAJOR, ldb=N, k_batch=kb)
    exp = g[f"{tag}_c_dec"].reshape(M, N)
    ulp = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    tiny = np.abs(bits_to_f32(got, capi.BF16) - bits_to_f32(exp, capi.BF16)) <= 1e-3
    assert (ulp[~tiny] <= 1).all(), ulp[~tiny].max()
    # the same weights K-contiguous
    q = np.ascontiguousarray(g[f"{tag}_brow"].T)
    got = run_fpaintb(g[f"{tag}_a_int"], q, None, M, N, K, a_dtype=capi.BF16, b_dtype=capi.I8, k_batch=kb)
    np.testing.assert_array_equal(got, g[f"{tag}_c_int"].reshape(M, N))


def test_reference_splitk_test_shapes_through_ck_profiling_loop():
    """examples/_ref_build/ck_test_gemm_splitk_weight_quant_dropin: every shape of CK's split-K test x KBatch {1, 2, 3, 5, 8} x
    four layouts x two activation types through CK's unmodified profile_gemm_splitk_impl.hpp over the gfx950 registration:
    no wrong answer, and (as the build stands) no refusal either."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "_ref_build",
                       "ck_test_gemm_splitk_weight_quant_dropin")
    assert os.path.exists(exe), "examples/_ref_build/ck_test_gemm_splitk_weight_quant_dropin is not built (make -C examples dropin)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ", 0 wrong" in r.stdout and "WRONG" not in r.stdout
    assert " 0 refused" in r.stdout
