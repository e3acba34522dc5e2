"""Multi-rank split-K host logic on CPU: two gloo ranks partition K, each fills its fp32 partial
(the per-rank GEMM is injected -- here the CPU checker stands in for the GPU kernel), the partials
are summed with torch.distributed and rank 0 finalizes.  Covers partition(), reduce_partials() and
splitk_step() of composable_kernel_amd/splitk.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from composable_kernel_amd import splitk


def test_partition_rules():
    # KRead = ceil(K/(G*64))*64; the last slice takes the remainder; trailing slices may be empty
    sl = splitk.partition(65536, 8)
    assert [s.length for s in sl] == [8192] * 8 and sl[3].begin == 3 * 8192
    sl = splitk.partition(4096, 5)
    assert [s.length for s in sl] == [832, 832, 832, 832, 768]
    sl = splitk.partition(128, 4)
    assert [s.length for s in sl] == [64, 64, 0, 0]
    assert sum(s.length for s in splitk.partition(437 * 64, 3)) == 437 * 64
    for s in splitk.partition(1024, 3):
        assert s.begin % 64 == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, M, N, K, a, b, s, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding
    chk = binding.load_port()

    def compute_partial(sl, partial):
        if sl.length == 0:
            partial.zero_()
        else:
            partial.copy_(torch.from_numpy(chk.fpAintB_partial(a, b, s, M, N, K, K, sl.begin, sl.end, nthreads=2)))

    out = {}

    def finalize(summed, c_rows, row0):
        c_rows.copy_(torch.from_numpy(chk.splitk_reduce_f16(summed.numpy()[None]).reshape(summed.shape).view(np.int16)))
        out["row0"] = row0

    partial = torch.empty((M, N), dtype=torch.float32)
    c = torch.zeros((M, N), dtype=torch.int16)
    splitk.splitk_step(compute_partial, finalize, partial, c, K, strategy="reduce")
    if rank == 0:
        q.put((c.numpy().view(np.uint16).copy(), out.get("row0")))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K", [256, 128])
def test_two_rank_splitk_matches_single_pass(port, K):
    rng = np.random.default_rng(K)
    M, N = 16, 24
    a = np.round(rng.uniform(-2, 2, (M, K))).astype(np.float16).view(np.uint16)
    b = rng.integers(-8, 8, (N, K)).astype(np.int8)
    s = rng.integers(1, 3, N).astype(np.float16).view(np.uint16)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_no = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port_no, M, N, K, a, b, s, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, row0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = port.fpAintB_gemm_f16(a, b, s, M, N, K, K, K, N).reshape(M, N)
    np.testing.assert_array_equal(got, exp)   # integer-valued inputs: exact
    assert row0 == 0


def _worker_gemm_step(rank, world, port, M, N, K, a, b, s, q):
    """The path bench.py --gpus N runs (splitk_gemm_step: reduce-scatter, row-slice finalize with its row offset,
    all-gather), with the CPU checker standing in for the per-rank kernels."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding
    chk = binding.load_port()
    seen = {}

    def compute_partial(sl, partial):
        seen["slice"] = (sl.begin, sl.end)
        if sl.length == 0:
            partial.zero_()
        else:
            partial.copy_(torch.from_numpy(chk.fpAintB_partial(a, b, s, M, N, K, K, sl.begin, sl.end, nthreads=2)))

    def finalize(summed, c_rows, row0):
        seen["row0"], seen["rows"] = row0, summed.shape[0]
        c_rows.copy_(torch.from_numpy(chk.splitk_reduce_f16(summed.numpy()[None]).reshape(summed.shape).view(np.int16)))

    partial = torch.empty((M, N), dtype=torch.float32)
    c = torch.full((M, N), 0x7BFF, dtype=torch.int16)
    splitk.splitk_gemm_step(compute_partial, finalize, partial, c, K)
    q.put((rank, c.numpy().view(np.uint16).copy(), seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,K", [(2, 256), (3, 256), (3, 128)])
def test_gemm_step_reduce_scatter_rows_gather(port, world, K):
    """World sizes 2 and 3; K = 128 on three ranks leaves the last rank an empty slice (zero partial).  Every rank must
    end with the complete C, equal to the single-pass result (integer-valued inputs: exact)."""
    rng = np.random.default_rng(K + world)
    M, N = 24, 16
    a = np.round(rng.uniform(-2, 2, (M, K))).astype(np.float16).view(np.uint16)
    b = rng.integers(-8, 8, (N, K)).astype(np.int8)
    s = rng.integers(1, 3, N).astype(np.float16).view(np.uint16)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_no = _free_port()
    procs = [ctx.Process(target=_worker_gemm_step, args=(r, world, port_no, M, N, K, a, b, s, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = port.fpAintB_gemm_f16(a, b, s, M, N, K, K, K, N).reshape(M, N)
    slices = splitk.partition(K, world)
    for rank, got, seen in results:
        np.testing.assert_array_equal(got, exp)
        assert seen["row0"] == rank * (M // world) and seen["rows"] == M // world
        assert seen["slice"] == (slices[rank].begin, slices[rank].end)
    if K == 128 and world == 3:
        assert slices[2].length == 0
