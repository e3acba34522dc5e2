"""Multi-GPU split-K: K is partitioned across the ranks of one node, every rank runs the fpAintB
kernel on its slice into an fp32 partial, the partials are summed over RCCL (xGMI) and the
epilogue/convert kernel produces C.

The reference has no communication code; its split-K is intra-GPU (blockIdx.z = k-batch, fp32
workspace [KBatch, M, N], then a reduce kernel:
include/ck/tensor_operation/gpu/grid/gridwise_gemm_xdl_cshuffle_v3.hpp:154-157,560-604 and
device/impl/device_gemm_xdl_cshuffle_v3r1.hpp:209-270).  This module keeps that shape -- slice,
fp32 partial, sum in ascending slice order, convert -- with the sum done by a collective.

One process per GPU; ``torch.distributed`` supplies the process group ("nccl" is RCCL on ROCm;
"gloo" on CPU for tests, with the per-rank GEMM injected by the caller).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable

import torch
import torch.distributed as dist

K_GRANULE = 64  # K extent of one kernel tile: slices start on multiples of it


def kread(K: int, parts: int, granule: int = K_GRANULE) -> int:
    """ceil(K / (parts * granule)) * granule (gridwise_gemm_xdl_cshuffle_v3.hpp:192-197)."""
    per = parts * granule
    return (K + per - 1) // per * granule


@dataclass(frozen=True)
class KSlice:
    rank: int
    begin: int
    end: int

    @property
    def length(self) -> int:
        return max(0, self.end - self.begin)


def partition(K: int, world: int, granule: int = K_GRANULE) -> list[KSlice]:
    """Slice g = [g*KRead, min(K, (g+1)*KRead)); the last non-empty slice takes the remainder
    (gridwise_gemm_xdl_cshuffle_v3.hpp:583-590).  Trailing ranks may be empty for small K."""
    kr = kread(K, world, granule)
    return [KSlice(g, min(K, g * kr), min(K, (g + 1) * kr)) for g in range(world)]


def reduce_partials(partial: torch.Tensor, strategy: str = "auto", group=None) -> torch.Tensor | None:
    """Sum the per-rank fp32 partials [M, N].

    "reduce"          one root-directed reduce; rank 0 gets the full [M, N] sum, others None.
    "reduce_scatter"  each rank gets rows [r*M/G, (r+1)*M/G) of the sum (M % G == 0): every rank
                      sends 1/G of its partial to every peer at once, which is what point-to-point
                      xGMI links favour over a root-bound ring.
    "auto"            reduce_scatter on RCCL when M divides evenly, else reduce.
    """
    world = dist.get_world_size(group)
    if world == 1:
        return partial
    backend = dist.get_backend(group)
    if strategy == "auto":
        strategy = "reduce_scatter" if backend == "nccl" and partial.shape[0] % world == 0 else "reduce"
    if strategy == "reduce":
        dist.reduce(partial, dst=0, op=dist.ReduceOp.SUM, group=group)
        return partial if dist.get_rank(group) == 0 else None
    if strategy == "reduce_scatter":
        M = partial.shape[0]
        assert M % world == 0, "reduce_scatter needs M divisible by the world size"
        rows = M // world
        if backend != "nccl":
            # gloo has no reduce-scatter: sum everywhere, keep this rank's rows (same result, CPU test path only)
            dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
            r = dist.get_rank(group)
            return partial[r * rows:(r + 1) * rows].clone()
        out = torch.empty((rows,) + tuple(partial.shape[1:]), dtype=partial.dtype, device=partial.device)
        dist.reduce_scatter_tensor(out, partial, op=dist.ReduceOp.SUM, group=group)
        return out
    raise ValueError(f"unknown strategy {strategy!r}")


def splitk_step(compute_partial: Callable[[KSlice, torch.Tensor], None],
                finalize: Callable[[torch.Tensor, torch.Tensor, int], None],
                partial: torch.Tensor, c_rows: torch.Tensor, K: int, strategy: str = "auto",
                group=None) -> None:
    """One split-K GEMM over the process group.

    compute_partial(slice, partial)   fill ``partial`` [M, N] fp32 with this rank's K slice
                                      (zero it for an empty slice)
    finalize(sum_rows, c_rows, row0)  epilogue + convert of summed rows starting at row ``row0``
    partial                           preallocated [M, N] fp32 scratch
    c_rows                            where this rank's share of C goes: [M, N] on rank 0 for
                                      "reduce", [M/G, N] on every rank for "reduce_scatter"
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sl = partition(K, world)[rank]
    compute_partial(sl, partial)
    if world == 1:
        finalize(partial, c_rows, 0)
        return
    summed = reduce_partials(partial, strategy, group)
    if summed is None:
        return
    row0 = 0 if summed.shape[0] == partial.shape[0] else rank * summed.shape[0]
    finalize(summed, c_rows, row0)


def gather_rows(c_full: torch.Tensor, rows: int, group=None) -> None:
    """All-gather in place: every rank has written rows [r*rows, (r+1)*rows) of the contiguous ``c_full`` [M, N]."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    r = dist.get_rank(group)
    mine = c_full[r * rows:(r + 1) * rows]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(c_full, mine.clone(), group=group)
    else:
        # gloo (CPU test path): gather the rows as bytes, whatever the 16-bit element type is
        mine_b = mine.contiguous().view(torch.uint8)
        parts = [torch.empty_like(mine_b) for _ in range(world)]
        dist.all_gather(parts, mine_b.clone(), group=group)
        for g, part in enumerate(parts):
            c_full[g * rows:(g + 1) * rows].copy_(part.view(c_full.dtype).view(mine.shape))


def splitk_gemm_step(compute_partial: Callable[[KSlice, torch.Tensor], None],
                     finalize: Callable[[torch.Tensor, torch.Tensor, int], None],
                     partial: torch.Tensor, c_full: torch.Tensor, K: int, group=None) -> None:
    """The multi-GPU split-K GEMM as ``bench.py --gpus N`` times it, C complete on every rank afterwards:

        slice GEMM -> fp32 partial [M, N]   (compute_partial; zero it for an empty slice)
        reduce-scatter of the partials      (rank r keeps rows [r*M/G, (r+1)*M/G) of the sum)
        finalize those rows                 (finalize(sum_rows, c_full[rows of r], row0): epilogue + convert)
        all-gather of the finished rows     (in place in ``c_full`` [M, N], contiguous)

    M must be divisible by the world size.  One rank: GEMM + finalize only.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    M = partial.shape[0]
    assert M % world == 0, "M must be divisible by the world size"
    rows = M // world
    compute_partial(partition(K, world)[rank], partial)
    summed = partial if world == 1 else reduce_partials(partial, "reduce_scatter", group)
    finalize(summed, c_full[rank * rows:(rank + 1) * rows], rank * rows)
    if world > 1:
        gather_rows(c_full, rows, group)
