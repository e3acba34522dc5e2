"""PyTorch-ROCm front end of the weight-quantized GEMM: a linear layer whose weights live on the GPU in the
engine's packed formats, and a ``torch.library`` custom operator around the C ABI launch.

PyTorch supplies device memory and the stream; the arithmetic is the HIP kernels behind ``wqgemm_launch``
(there is no eager fallback: without the library or a GPU the operator raises).
"""
from __future__ import annotations

import torch

from . import capi
from .weights import PackedWeights

_DT = {torch.float16: capi.F16, torch.bfloat16: capi.BF16}
_EPILOGUES = {"none": capi.EPI_PASSTHROUGH, "fastgelu": capi.EPI_FASTGELU, "add": capi.EPI_ADD,
              "add_fastgelu": capi.EPI_ADD_FASTGELU}


@torch.library.custom_op("wqgemm::fpaintb_linear", mutates_args=())
def fpaintb_linear(x: torch.Tensor, weight: torch.Tensor, scale: torch.Tensor, bits: int, epilogue: str,
                   bias: torch.Tensor | None, layout: str = "packed") -> torch.Tensor:
    """y[m, n] = epilogue( sum_k x[m, k] * dequant(weight[n, k], scale[n]) [, bias[n]] ).

    layout = "preshuffled": the weight bytes are in the order of wqgemm_preshuffle_int4 / _int8; needs N a multiple of 32
    and K of 64, and raises where no matrix-core kernel can take the problem.

    x (M, K) fp16 / bf16, row-major; weight uint8 (N, K/2) packed int4 (bits = 4) or int8 (N, K) (bits = 8);
    scale and bias (N,) of x's dtype; epilogue in {"none", "fastgelu", "add", "add_fastgelu"}.
    Group-wise scales: scale of shape (G, N), one row per group of K / G consecutive k."""
    if not x.is_cuda:
        raise RuntimeError("wqgemm::fpaintb_linear runs on the GPU only")
    if x.dtype not in _DT or scale.dtype != x.dtype:
        raise TypeError("x and scale must both be float16 or both bfloat16")
    if epilogue not in _EPILOGUES:
        raise ValueError(f"unknown epilogue {epilogue!r}")
    if (bias is None) != (epilogue in ("none", "fastgelu")):
        raise ValueError("bias goes with the add / add_fastgelu epilogues only")
    if bits not in (4, 8):
        raise ValueError(f"bits must be 4 (packed int4) or 8 (int8), got {bits}")
    if layout not in ("packed", "preshuffled"):
        raise ValueError(f"layout must be 'packed' or 'preshuffled', got {layout!r}")
    if x.dim() != 2 or scale.dim() not in (1, 2):
        raise ValueError("x must be (M, K) and scale (N,) or (groups, N)")
    M, K = x.shape
    N = scale.shape[-1]
    group_k = 0
    if scale.dim() == 2:
        if scale.shape[0] == 0 or K % scale.shape[0] != 0:
            raise ValueError(f"K = {K} is not a whole number of the {scale.shape[0]} scale groups")
        group_k = K // scale.shape[0]
    # the kernels read weight / scale / bias through raw pointers: shapes, dtypes, devices and contiguity are checked here
    want = (N, K // 2) if bits == 4 else (N, K)
    if bits == 4 and K % 8 != 0:
        raise ValueError("packed int4 weights need K to be a multiple of 8")
    if tuple(weight.shape) != want or weight.dtype not in (torch.uint8, torch.int8):
        raise ValueError(f"weight must be an 8-bit tensor of shape {want} for bits = {bits}, got {tuple(weight.shape)} {weight.dtype}")
    for name, t in (("weight", weight), ("scale", scale), ("bias", bias)):
        if t is None:
            continue
        if t.device != x.device:
            raise ValueError(f"{name} is on {t.device}, x on {x.device}")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
    if bias is not None and (tuple(bias.shape) != (N,) or bias.dtype != x.dtype):
        raise ValueError(f"bias must have shape ({N},) and x's dtype {x.dtype}, got {tuple(bias.shape)} {bias.dtype}")
    x = x.contiguous()
    # any M runs as it is: small-M kernel up to 64 rows, the tile kernels with a partial last row tile beyond
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    p = capi.fpAintB_problem(x.data_ptr(), weight.data_ptr(), scale.data_ptr(), y.data_ptr(), M, N, K,
                             a_dtype=_DT[x.dtype],
                             b_dtype=((capi.I4_PRESHUFFLED, capi.I8_PRESHUFFLED) if layout == "preshuffled"
                                      else (capi.I4_PACKED, capi.I8))[bits == 8],
                             epilogue=_EPILOGUES[epilogue], d0=bias.data_ptr() if bias is not None else None,
                             scale_group_k=group_k)
    if not capi.is_supported(p):
        raise RuntimeError("wqgemm::fpaintb_linear: " + capi.lib.wqgemm_last_error().decode())
    nbytes = capi.workspace_bytes(p)  # optional scratch (the planner's split-K partials)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    with torch.cuda.device(x.device):
        capi.launch(p, ws.data_ptr() if ws is not None else None, torch.cuda.current_stream().cuda_stream)
    return y


@fpaintb_linear.register_fake
def _(x, weight, scale, bits, epilogue, bias, layout="packed"):
    return x.new_empty((x.shape[0], scale.shape[-1]))


class WqLinear(torch.nn.Module):
    """Inference-only linear layer over packed weights (see ``weights.pack`` / ``weights.load_packed``)."""

    def __init__(self, pw: PackedWeights, bias: torch.Tensor | None = None, epilogue: str = "none", device="cuda"):
        super().__init__()
        dt = torch.float16 if pw.scale_dtype == "f16" else torch.bfloat16
        self.bits, self.in_features, self.out_features, self.layout = pw.bits, pw.K, pw.N, pw.layout
        self.register_buffer("weight", torch.from_numpy(pw.weight.copy()).to(device))
        self.register_buffer("scale", torch.from_numpy(pw.scale.view("int16").copy()).view(dt).to(device))
        self.register_buffer("bias", None if bias is None else bias.to(device=device, dtype=dt))
        self.epilogue = ("add" if epilogue == "none" else "add_fastgelu") if bias is not None else epilogue

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        y = fpaintb_linear(x.reshape(-1, self.in_features), self.weight, self.scale, self.bits, self.epilogue, self.bias,
                           self.layout)
        return y.reshape(*lead, self.out_features)
