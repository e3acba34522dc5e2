"""The multi-GPU split-K entry points on the one-GPU box: everything that can run with one rank runs through a REAL
RCCL communicator (wqgemm_splitk_multi_step issues ncclReduceScatter and ncclAllGather on it), and asking for more GPUs
than the node has is a clear refusal -- from `bench.py --gpus N` started by hand (it starts its own ranks) and from the
C++ driver (it forks one process per GPU before anything touches the GPU runtime)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from composable_kernel_amd import capi, rccl
from gpu_util import dev, f16_bits, host_u16, pack_on_device

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_splitk_multi_step_through_a_real_rccl_communicator(port):
    """One rank, communicator made with ncclGetUniqueId + ncclCommInitRank: the reduce-scatter and the in-place
    all-gather run, and C equals the single-pass kernel's output bit for bit -- for a full-size slice of config 4
    (4096 x 4096 x 8192, pre-shuffled weights) and for a small fp8 problem with a per-row epilogue operand."""
    comm = rccl.comm_init_rank(1, rccl.get_unique_id(), 0)
    try:
        st = torch.cuda.current_stream().cuda_stream
        g = torch.Generator(device="cuda")
        g.manual_seed(8)
        M = N = 4096
        K = 8192
        a = (torch.rand((M, K), generator=g, device="cuda") * 2 - 1).half()
        q = torch.randint(-8, 8, (N, K), generator=g, device="cuda", dtype=torch.int8)
        s = (torch.rand((N,), generator=g, device="cuda") * (2.0 ** -4 - 2.0 ** -8) + 2.0 ** -8).half()
        bp = torch.empty((N, K // 2), dtype=torch.uint8, device="cuda")
        capi.pack_int4(q.data_ptr(), K, K, N, bp.data_ptr(), K // 2, st)
        pre = torch.empty_like(bp)
        capi.preshuffle_int4(bp.data_ptr(), K // 2, K, N, pre.data_ptr(), K, st)
        c_ref = torch.zeros((M, N), dtype=torch.int16, device="cuda")
        c = torch.zeros_like(c_ref)
        p_ref = capi.fpAintB_problem(a.data_ptr(), pre.data_ptr(), s.data_ptr(), c_ref.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)
        capi.launch(p_ref, None, st)
        p = capi.fpAintB_problem(a.data_ptr(), pre.data_ptr(), s.data_ptr(), c.data_ptr(), M, N, K, b_dtype=capi.I4_PRESHUFFLED)
        ws = torch.empty(capi.splitk_multi_workspace_bytes(p, 1), dtype=torch.uint8, device="cuda")
        for _ in range(2):    # twice: the workspace is reused as it is left
            capi.splitk_multi_step(p, comm, 0, 1, ws.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(c, c_ref)
        rows = slice(77, 81)
        exp = port.fpAintB_gemm_f16(a[rows].cpu().numpy().view(np.uint16), q.cpu().numpy(), s.cpu().numpy().view(np.uint16), 4, N, K,
                                    K, K, N).reshape(4, N)
        got = host_u16(c[rows]).view(np.float16).astype(np.float32)
        err = np.abs(got - exp.view(np.float16).astype(np.float32))
        assert (err <= 1.5e-3 + 1.5e-3 * np.abs(got)).all()
        del a, q, bp, pre, c, c_ref, ws
        # fp8 activations, MultiplyMultiply with d1 per row, through the same communicator
        rng = np.random.default_rng(9)
        M, N, K = 512, 512, 1024
        a8 = dev(rng.choice(np.array([0x00, 0x38, 0x40, 0xB8, 0xC0], np.uint8), (M, K)))
        b_t = pack_on_device(rng.integers(-8, 8, (N, K)).astype(np.int8), K, N)
        d0 = dev(rng.uniform(0.5, 1.0, N).astype(np.float32))
        d1 = dev(rng.uniform(0.5, 1.0, M).astype(np.float32))
        kw = dict(a_dtype=capi.F8_E4M3FN, c_dtype=capi.F16, epilogue=capi.EPI_MULTIPLY_MULTIPLY, d0=d0.data_ptr(), d1=d1.data_ptr())
        c_ref = torch.zeros((M, N), dtype=torch.int16, device="cuda")
        c = torch.zeros_like(c_ref)
        capi.launch(capi.fpAintB_problem(a8.data_ptr(), b_t.data_ptr(), None, c_ref.data_ptr(), M, N, K, **kw), None, st)
        p2 = capi.fpAintB_problem(a8.data_ptr(), b_t.data_ptr(), None, c.data_ptr(), M, N, K, **kw)
        ws2 = torch.empty(capi.splitk_multi_workspace_bytes(p2, 1), dtype=torch.uint8, device="cuda")
        capi.splitk_multi_step(p2, comm, 0, 1, ws2.data_ptr(), st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(host_u16(c), host_u16(c_ref))
    finally:
        torch.cuda.synchronize()
        rccl.comm_destroy(comm)


def test_bench_started_by_hand_refuses_more_gpus_than_the_node_has():
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, env=_clean_env(), timeout=300)
    assert r.returncode != 0
    assert f"needs {have + 1} GPUs, found {have}" in r.stderr and r.stdout.strip() == ""


def test_bench_split_k_flow_starts_its_own_rank_and_prints_one_line():
    """`bench.py --config C4 --gpus 1`, no launcher and no RANK in the environment: the parent starts one rank through
    torch.distributed.run and passes its JSON line on.  The step timed is the C ABI's, over a real communicator."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C4", "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--splitk-k", "8192", "--ramp-launches", "0"], capture_output=True, text=True, env=_clean_env(), timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["K"] == 8192 and out["value"] > 0
    assert out["exchange"]["timed"] == "c_abi_rccl" and out["exchange"]["communicator_error"] is None
    chk = out["splitk_check"]
    assert chk["exact_match"] and chk["ranks_agree"]
    assert chk["c_abi_rccl"]["exact_match"] and chk["torch_distributed"]["exact_match"]
    t = out["timing"]
    assert t["steps_only_ms_per_step"] <= out["ms_per_step"] + 1e-9


@pytest.mark.parametrize("shape", [(1024, 1024, 8192), (512, 768, 4160)])
def test_cpp_driver_one_rank(shape):
    """examples/example_splitk_multi_gpu with one rank: fork before any GPU call, unique id over pipes, ncclCommInitRank,
    wqgemm_splitk_multi_step, exact-integer check of the result."""
    exe = os.path.join(ROOT, "examples", "example_splitk_multi_gpu")
    assert os.path.exists(exe), "examples/example_splitk_multi_gpu is not built: run `make -C examples`"
    r = subprocess.run([exe, "1", *map(str, shape), "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verification: PASS" in r.stdout and "0 wrong" in r.stdout and "Perf:" in r.stdout


def test_cpp_driver_refuses_more_ranks_than_gpus():
    exe = os.path.join(ROOT, "examples", "example_splitk_multi_gpu")
    assert os.path.exists(exe)
    have = torch.cuda.device_count()
    r = subprocess.run([exe, str(have + 1), "512", "512", "1024"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert f"needs {have + 1} GPUs, found {have}" in r.stderr
