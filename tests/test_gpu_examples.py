"""The C++ client-API facade on the GPU: drivers written against CK's DeviceGemm* interfaces.

* examples/example_*_gfx950: this tree's own drivers (facade headers only), each verifying the
  device result against its host reference exactly as CK's examples do.
* examples/_ref_build/example_fp16int8_gemm_dropin: CK's OWN example/64 source compiled unmodified
  with include/dropin overlaid in front of CK's headers (built only where the CK tree is mounted;
  the binary travels, the sources do not).  Its verification uses CK's own host
  ReferencefpAintBGemm and check_err, so a pass is the reference checking this engine.
"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


def _run(path, *args):
    # a missing binary is a failure, not a skip: the boundary suite must never pass by not running
    assert os.path.exists(path), (f"{os.path.relpath(path, ROOT)} is not built: run `make -C examples` (and, where the CK tree "
                                  "is mounted, `make -C examples dropin`) before the GPU tests")
    r = subprocess.run([path, *map(str, args)], capture_output=True, text=True, timeout=300)
    print(r.stdout[-2000:], r.stderr[-2000:])
    return r


@pytest.mark.parametrize("init", [1, 2])
def test_fpAintB_example_standalone(init):
    r = _run(os.path.join(EX, "example_fpAintB_gemm_gfx950"), 1, init, 1, 256, 384, 512, 512, 512, 384)
    assert r.returncode == 0 and "verification: PASS" in r.stdout
    assert "DeviceFpAintBGemm_Mfma_gfx950" in r.stdout


@pytest.mark.parametrize("shape", [(256, 384, 512, 512, 512, 384), (1024, 512, 1024, 1024, 1088, 512), (24, 256, 512, 512, 512, 256)])
def test_fpAintB_example_preshuffled_weights(shape):
    """The same driver with the weights stored in the pre-shuffled order (host re-layout, facade type tag
    ck::wq_int4x2_preshuffled_t): 128-tile, 256-tile (padded K pitch) and small-M kernels, CK's host reference + check_err."""
    r = _run(os.path.join(EX, "example_fpAintB_gemm_gfx950"), 1, 2, 1, *shape, 1)
    assert r.returncode == 0 and "verification: PASS" in r.stdout
    assert "I4_PRESHUFFLED" in r.stdout


def test_fpAintB_example_ragged_shape_uses_generic_kernel():
    r = _run(os.path.join(EX, "example_fpAintB_gemm_gfx950"), 1, 2, 0, 100, 72, 200, 200, 200, 72)
    assert r.returncode == 0 and "verification: PASS" in r.stdout


def test_int8_quantization_example():
    r = _run(os.path.join(EX, "example_gemm_quantization_int8_gfx950"), 1, 512)
    assert r.returncode == 0 and "verification: PASS" in r.stdout


def test_int8_bias_relu_quantization_example():
    r = _run(os.path.join(EX, "example_gemm_bias_relu_quantization_int8_gfx950"), 1, 512)
    assert r.returncode == 0 and "verification: PASS" in r.stdout
    assert "DeviceGemmBiasQuantization_gfx950" in r.stdout


@pytest.mark.parametrize("init,shape", [(1, (512, 768, 1024)), (2, (512, 768, 1024)), (1, (100, 256, 256))])
def test_bf16_i8_convert_rtn_example(init, shape):
    """bf16 x int8 with ConvertBF16RTN as the C operation through DeviceGemmMultiABD_gfx950: bit-exact against the host
    ReferenceGemm on the half-integer set (init 1), within one bf16 step (rtol 1e-2) on decimal values (init 2)."""
    r = _run(os.path.join(EX, "example_gemm_bf16_i8_rtn_gfx950"), 1, init, 1, *shape)
    assert r.returncode == 0 and "verification: PASS" in r.stdout
    assert "Perf:" in r.stdout


@pytest.mark.parametrize("name,op", [("gemm_xdl_quantization_int8", "DeviceGemmQuantization_gfx950"),
                                     ("gemm_xdl_bias_relu_quantization_int8", "DeviceGemmBiasQuantization_gfx950")])
def test_reference_example14_sources_dropin(name, op):
    """CK's example/14_gemm_quantization XDL sources compiled unmodified (fixed 1024^3 problem): the binary returns 0
    only if ck::utils::check_err(device, CK's host reference) passed with the default int8 tolerance of 0."""
    r = _run(os.path.join(EX, "_ref_build", f"example14_{name}_dropin"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Perf:" in r.stdout and op in r.stdout


@pytest.mark.parametrize("init", [1, 2])
def test_reference_example64_source_dropin(init):
    """CK's example/64 binary (argv: verify init time M N K StrideA StrideB StrideC) returns 0
    only if ck::utils::check_err(device, host reference) passed."""
    r = _run(os.path.join(EX, "_ref_build", "example_fp16int8_gemm_dropin"), 1, init, 1, 256, 384, 512, 512, 512, 384)
    assert r.returncode == 0, r.stderr
    assert "does not support this problem" not in r.stderr
    assert "Perf:" in r.stdout and "DeviceFpAintBGemm_Mfma_gfx950" in r.stdout


@pytest.mark.parametrize("kbatch", [1, 2, 3, 5, 8])
def test_reference_example35_splitk_source_dropin(kbatch):
    """CK's example/35_splitK_gemm/gemm_xdl_splitk_reduce_bf16A_i8B.cpp compiled unmodified over include/dropin
    (argv: verify init time M N K StrideA StrideB StrideC KBatch; bf16 A, row-major int8 B): the caller sizes and hands
    over the workspace (GetWorkSpaceSize / SetWorkSpacePointer) and the binary returns 0 only if
    ck::utils::check_err(device, CK's ReferenceGemm) passed.  KBatch values: the reference's split-K test set
    (test/gemm_split_k/test_gemm_splitk_util.hpp:36-87)."""
    r = _run(os.path.join(EX, "_ref_build", "example35_gemm_xdl_splitk_reduce_bf16A_i8B_dropin"),
             1, 2, 1 if kbatch == 2 else 0, 256, 1024, 512, 512, 1024, 1024, kbatch)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "does not support this problem" not in r.stderr and f"KBatch: {kbatch}" in r.stdout
    if kbatch == 2:
        assert "Perf:" in r.stdout and "DeviceGemmV2R1_WeightQuant_gfx950" in r.stdout


def test_reference_example35_default_arguments():
    r = _run(os.path.join(EX, "_ref_build", "example35_gemm_xdl_splitk_reduce_bf16A_i8B_dropin"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Perf:" in r.stdout and "DeviceGemmV2R1_WeightQuant_gfx950" in r.stdout


@pytest.mark.parametrize("dtype,layout,kbatch", [(7, 0, 0), (7, 1, 3), (8, 0, 5), (8, 1, 2)])
def test_reference_profiler_splitk_loop_dropin(dtype, layout, kbatch):
    """CK's OWN ckProfiler split-K loop (profile_gemm_splitk_impl.hpp, unmodified) instantiated for int8 weights: it
    finds the gfx950 operator through the factory header this engine adds, sweeps KBatch (0 = CK's list 1 .. 38) and
    verifies every run against ReferenceGemm with check_err's defaults."""
    r = _run(os.path.join(EX, "_ref_build", "ck_profile_gemm_splitk_weight_quant_dropin"),
             "gemm_splitk", dtype, layout, 1, 1, 0, 1, 256, 512, 1024, -1, -1, -1, kbatch)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "found 1 instances" in r.stdout and "Best Perf" in r.stdout
    assert "DeviceGemmSplitK_WeightQuant_gfx950" in r.stdout and "does not support" not in r.stdout
    if kbatch == 0:
        assert r.stdout.count("Perf:") >= 10


@pytest.mark.parametrize("dtype,init,extra", [(8, 1, ()), (9, 2, ()), (10, 1, ()), (10, 2, ()), (11, 2, ()),
                                              (10, 2, (2, 7, 64)), (8, 2, (1, 5, 64))])
def test_reference_profiler_multiply_multiply_loop_dropin(dtype, init, extra):
    """CK's OWN ckProfiler MultiplyMultiply loop (profile_gemm_multiply_multiply_impl.hpp, unmodified) instantiated for fp8
    activations x ck::int4_t weights (one value per byte, CK's int4 convention): CK generates and uploads the tensors --
    D0 and D1 as full (M, N) matrices, row- and column-major -- runs ReferenceGemm + MultiplyMultiply on the host and
    check_err; the gfx950 operator is found through the factory header this engine adds, packs the weights two per byte
    and runs the fp8 matrix-core kernel (E4M3FN bytes, data types 10 / 11) or the exact fp16-image route (FNUZ bytes,
    8 / 9).  `extra` = warm-up, iterations, rotating-buffer MB: the flush-cache StreamConfig of the loop's timed run."""
    M, N, K = 512, 768, 1024
    r = _run(os.path.join(EX, "_ref_build", "ck_profile_gemm_multiply_multiply_weight_quant_dropin"),
             "gemm_multiply_multiply", dtype, 1, 1, init, 0, 1, M, N, K, -1, -1, -1, -1, -1, *extra)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "found 1 instances" in r.stdout and "Best Perf" in r.stdout and "does not support" not in r.stdout
    assert "DeviceGemmMultiplyMultiply_Int4Values_gfx950<" + ("F8_E4M3FN" if dtype >= 10 else "F8_FNUZ") in r.stdout
    if extra:
        # 64 MB over A (0.5 MiB) + B (0.75 MiB, one byte per weight) + D0, D1 (1.5 MiB each): more copies than iterations
        assert f"rotating count: {extra[1]}" in r.stdout


@pytest.mark.parametrize("dtype,init,extra", [(9, 1, ()), (9, 2, ()), (8, 1, ()), (8, 2, ()), (10, 1, ()), (10, 2, ()), (10, 2, (2, 7, 64))])
def test_reference_profiler_ab_scale_loop_dropin(dtype, init, extra):
    """CK's OWN ckProfiler block-scale loop (profile_gemm_ab_scale_impl.hpp, unmodified) instantiated for fp8 activations x
    ck::int4_t weights with 128 x 128 x 128 scale blocks of both operands: CK generates and uploads operands and scale
    tensors, runs its host check (operands scaled to fp32, ReferenceGemm, convert) and check_err; the gfx950 operators are
    found through the factory header this engine adds, pack the weights two per byte and run the block-scaled fp8
    matrix-core kernel (E4M3FN bytes: one instance; FNUZ bytes: that kernel under the no-special-encodings promise next to
    the exact general-kernel instance).  (E4M3FN -> bf16, data type 11, is left out: for any A type other than ck::f8_t that
    loop checks bf16 outputs at check_err's default rtol 1e-3, a fifth of a bf16 rounding step, which only a kernel summing in
    the host's exact order can meet; for ck::f8_t it uses 5e-2.)"""
    M, N, K = 512, 768, 1024
    r = _run(os.path.join(EX, "_ref_build", "ck_profile_gemm_ab_scale_weight_quant_dropin"),
             "gemm_ab_scale", dtype, 1, 0, 1, init, 0, 1, M, N, K, -1, -1, -1, *extra)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    n_inst = 1 if dtype >= 10 else 2
    assert f"found {n_inst} instances" in r.stdout and "Best Perf" in r.stdout and "does not support" not in r.stdout
    assert r.stdout.count("Perf:") == n_inst
    assert "DeviceGemmMultiD_ABScale_Int4Values_gfx950<" + ("F8_E4M3FN" if dtype >= 10 else "F8_FNUZ") in r.stdout
    if extra:
        assert f"rotating count: {extra[1]}" in r.stdout


@pytest.mark.parametrize("fmt,gelu,init", [(0, 0, 1), (1, 0, 1), (1, 1, 2), (2, 0, 1), (2, 1, 2)])
def test_fp8_int4_multiply_multiply_example(fmt, gelu, init):
    """example/65's flow (fp8 A, D0 per column, D1 per row, host ReferenceGemm + check_err): FNUZ bytes through
    the fp16 image in the workspace (fmt 0) and on the fp8 matrix core (fmt 1), E4M3FN bytes on the matrix core
    (fmt 2)."""
    size = (512, 768, 1024)
    r = _run(os.path.join(EX, "example_gemm_multiply_multiply_fp8_int4_gfx950"), 1, init, 1, *size, fmt, gelu)
    assert r.returncode == 0 and "verification: PASS" in r.stdout, r.stdout + r.stderr
    assert ("16x16x128_f8f6f4" in r.stdout) == (fmt != 0) and "Mfma" in r.stdout


@pytest.mark.parametrize("fmt,init", [(1, 1), (2, 2)])
def test_fp8_int4_ab_scale_example_per_token(fmt, init):
    """The same flow with one activation scale per token and 128 k (ScaleBlockM = 1): the row-scale form of the kernel."""
    r = _run(os.path.join(EX, "example_gemm_ab_scale_fp8_int4_gfx950"), 1, init, 1, 640, 768, 1024, fmt, 1)
    assert r.returncode == 0 and "verification: PASS" in r.stdout, r.stdout + r.stderr
    assert "RowScaleA" in r.stdout, r.stdout


@pytest.mark.parametrize("fmt,init", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_fp8_int4_ab_scale_example(fmt, init):
    """example/65's block-scaled flow (fp8 A with a scale per 128 x 128 block, weights with a scale per 128 x 128 block,
    host check on operands scaled to fp32): bit-exact on integer values with power-of-two scales (init 1), within the
    example's 5e-2 otherwise; both fp8 encodings, a partial last row tile."""
    r = _run(os.path.join(EX, "example_gemm_ab_scale_fp8_int4_gfx950"), 1, init, 1, 640, 768, 1024, fmt)
    assert r.returncode == 0 and "verification: PASS" in r.stdout, r.stdout + r.stderr
    assert "16x16x128_f8f6f4" in r.stdout, r.stdout


@pytest.mark.parametrize("args", [
    ("gemm_ab_scale", 9, 1, 0, 1, 2, 0, 1, 640, 512, 1024, -1, -1, -1),
    ("gemm_ab_scale", 8, 1, 0, 1, 1, 0, 1, 256, 256, 384, -1, -1, -1),
    ("gemm_ab_scale", 10, 1, 0, 1, 2, 0, 1, 512, 768, 512, -1, -1, -1),
    ("gemm_ab_scale", 11, 1, 0, 1, 1, 0, 1, 300, 256, 256, -1, -1, -1, 2, 7, 16),
])
def test_profiler_cli_ab_scale(args):
    """The gemm_ab_scale operation of the ckProfiler-style command line (argv positions of profile_gemm_ab_scale.cpp: the
    scale-block-tile selector at arg4), verified by the reference's host check at its fp8 tolerances."""
    r = _run(os.path.join(EX, "example_wq_profiler"), *args)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verification: PASS" in r.stdout and "Best Perf for datatype" in r.stdout
    assert "ABScalePer128K" in r.stdout
    if len(args) == 17:
        assert "rotating count: 7" in r.stdout


def test_reference_client_example09_source_dropin():
    """CK's client_example/09_quantization/gemm_quantization.cpp, compiled unmodified, must find the gfx950
    operator through CK's own DeviceOperationInstanceFactory (the add_device_* symbols come from
    composable_kernel_amd/ck_instances/) and run it."""
    r = _run(os.path.join(EX, "_ref_build", "client_gemm_quantization_dropin"))
    assert r.returncode == 0, r.stderr
    assert "found 1 instances" in r.stdout
    assert "Best Perf" in r.stdout and "DeviceGemmQuantization_gfx950" in r.stdout and "Done" in r.stdout
    assert "does not support" not in r.stdout


@pytest.mark.parametrize("name", ["gemm_xdl_bf16_i8", "gemm_bias_xdl_bf16_i8", "gemm_bias_fastgelu_xdl_bf16_i8",
                                  "gemm_xdl_gelu_bf16_i8", "gemm_xdl_multiply_bf16_i8"])
def test_reference_client_example30_sources_dropin(name):
    """CK's client_example/30_gemm_bf16Aint8B sources, compiled unmodified, find the gfx950 bf16 x int8 operators
    through CK's own multi-ABD instance factory (add_device_* symbols from composable_kernel_amd/ck_instances/)."""
    r = _run(os.path.join(EX, "_ref_build", f"client30_{name}_dropin"))
    assert r.returncode == 0, r.stderr
    assert "found 1 instances" in r.stdout and "Best Perf" in r.stdout and "DeviceGemmMultiABD_gfx950" in r.stdout
    assert "does not support" not in r.stdout


@pytest.mark.parametrize("args", [
    ("gemm_fpAintB", 0, 1, 1, 2, 0, 1, 256, 256, 512, -1, -1, -1),
    ("gemm_fpAintB", 3, 1, 1, 2, 0, 1, 256, 384, 512, -1, -1, -1),
    ("gemm_fpAintB", 2, 1, 1, 1, 0, 0, 100, 72, 200, -1, -1, -1),
    ("gemm_fpAintB", 0, 1, 1, 1, 0, 1, 300, 16, 1024, -1, -1, -1),    # narrow N, ragged M: partial tiles + planner split-K
    ("gemm_fpAintB", 0, 1, 1, 2, 0, 1, 16, 2048, 8192, -1, -1, -1),   # small M: streaming kernel
    ("gemm_quantization", 3, 1, 1, 2, 0, 1, 300, 100, 512, -1, -1, -1),
    ("gemm_splitk", 0, 1, 1, 1, 0, 1, 256, 256, 2048, -1, -1, -1, 4),
    ("gemm_quantization", 3, 1, 1, 2, 0, 1, 512, 512, 512, -1, -1, -1),
    ("gemm_multiply_multiply", 8, 1, 1, 2, 0, 1, 256, 256, 512, -1, -1, 0, 0, -1),
    ("gemm_multiply_multiply", 9, 1, 1, 2, 0, 1, 256, 512, 512, -1, -1, 0, 0, -1),
    ("gemm_multiply_multiply", 10, 1, 1, 1, 0, 1, 512, 256, 512, -1, -1, 0, 0, -1),
    # arg16-18: warm-up cycles, iterations, memory for the rotating buffer in MB (the reference's flush-cache timing mode)
    ("gemm_multiply_multiply", 10, 1, 1, 2, 0, 1, 1024, 1024, 1024, -1, -1, 0, 0, -1, 2, 9, 16),
    # the reference's plain "gemm" operation name with data type 3 (int8), optional warm-up / iteration counts
    ("gemm", 3, 1, 1, 2, 0, 1, 512, 512, 512, -1, -1, -1),
    ("gemm", 3, 1, 1, 1, 0, 1, 300, 100, 512, -1, -1, -1, 2, 5),
])
def test_profiler_cli(args):
    """ckProfiler-style command line (operation name, data type, layout, verify, init, log, time, sizes, strides)."""
    r = _run(os.path.join(EX, "example_wq_profiler"), *args)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verification: PASS" in r.stdout
    if args[6]:
        assert "Best Perf for datatype" in r.stdout and "TFlops" in r.stdout
    if len(args) == 18:
        # 16 MB over (1 MiB A + 0.5 MiB B + 8 KiB D) -> 11 copies, capped by the 9 iterations
        assert "rotating count: 9" in r.stdout
