"""Host-side view of the HQQ on-disk weight format ("Format A") that AMQ's
pipeline produces (amq/amq_quantization_proxy.py:22-42 -> HQQLinear state).

This is input-format plumbing for the drop-in path (torch integer ops only):
  * ``HQQWeights``  -- what an HQQLinear carries: W_q, meta['scale'|'zero'|...]
  * ``pack_rows``   -- HQQ BitPack row-chunk packing (hqq/core/bitpack.py:24-110)
  * ``quantize_rtn``-- a plain round-to-nearest min/max quantizer emitting the
    same format (hqq/core/quantize.py:76-180 without the proximal optimizer),
    used to make synthetic checkpoints of real shapes: there is no network, so
    real AMQ/HQQ checkpoints are optional inputs.
  * ``from_hqq_layer`` -- duck-typed adaptor for the reference's HQQLinear.
"""
from dataclasses import dataclass
from typing import Optional

import torch

GROUP = 128
PACKING = {4: "4bit_u8", 3: "3bit_32", 2: "2bit_u8"}


@dataclass
class HQQWeights:
    W_q: torch.Tensor            # uint8 [R*bits/8, G] (4/2 bit) or int32 [ceil(R/10), G] (3 bit)
    scale: torch.Tensor          # fp16 [R, 1]  dequant multiplier
    zero: torch.Tensor           # fp16 [R, 1]
    nbits: int
    shape: tuple                 # (N, K)
    group_size: int = GROUP
    bias: Optional[torch.Tensor] = None
    name: Optional[str] = None

    @property
    def meta(self):
        """dict shaped like HQQLinear.meta (quantize.py:156-166)"""
        return {"nbits": self.nbits, "group_size": self.group_size, "shape": tuple(self.shape),
                "scale": self.scale, "zero": self.zero, "axis": 1, "packing": PACKING[self.nbits],
                "view_as_float": False}

    def to(self, device):
        return HQQWeights(self.W_q.to(device), self.scale.to(device), self.zero.to(device), self.nbits,
                          tuple(self.shape), self.group_size,
                          None if self.bias is None else self.bias.to(device), self.name)


def pack_rows(wg: torch.Tensor, nbits: int) -> torch.Tensor:
    """[R, G] integer groups -> HQQ W_q.  Chunk c of the rows goes to a fixed
    bit offset of every output word (bitpack.py:24-29, 43-53, 69-92)."""
    if nbits == 4:
        wg = wg.to(torch.uint8)
        step = wg.shape[0] // 2
        return (wg[:step] << 4) | wg[step:2 * step]
    if nbits == 2:
        wg = wg.to(torch.uint8)
        s = wg.shape[0] // 4
        return (wg[:s] << 6) | (wg[s:2 * s] << 4) | (wg[2 * s:3 * s] << 2) | wg[3 * s:4 * s]
    if nbits == 3:
        rows = -(-wg.shape[0] // 10) * 10
        pad = torch.zeros(rows, wg.shape[1], dtype=torch.int32, device=wg.device)
        pad[:wg.shape[0]] = wg.to(torch.int32)
        s = rows // 10
        out = torch.zeros(s, wg.shape[1], dtype=torch.int32, device=wg.device)
        for c in range(10):
            out |= pad[c * s:(c + 1) * s] << (27 - 3 * c)
        return out
    raise ValueError(f"nbits must be 2, 3 or 4 (got {nbits})")


def quantize_rtn(W: torch.Tensor, nbits: int, group_size: int = GROUP, bias=None, name=None) -> HQQWeights:
    """Min/max round-to-nearest quantization in HQQ's format and conventions
    (axis=1 grouping, inverted scale stored, fractional fp16 zero:
    quantize.py:106-155).  Not the HQQ optimizer -- accuracy is irrelevant to
    the speed path, only the format and value ranges matter."""
    if group_size not in (64, 32) and (group_size < GROUP or group_size % GROUP):
        raise ValueError("group size must be 32, 64 or a multiple of 128")
    n, k = W.shape
    wg = W.float().reshape(-1, group_size)
    mn = wg.min(dim=1, keepdim=True)[0]
    mx = wg.max(dim=1, keepdim=True)[0]
    maxv = float(2 ** nbits - 1)
    denom = mx - mn
    scale = maxv / denom
    scale = torch.where(denom.abs() <= 1e-4, torch.ones_like(scale), scale).clamp(max=2e4)
    zero = -mn * scale
    q = (wg * scale + zero).round().clamp(0, maxv)
    return HQQWeights(pack_rows(q, nbits), (1.0 / scale).to(torch.float16), zero.to(torch.float16),
                      nbits, (n, k), group_size, bias, name)


def random_hqq(n, k, nbits, seed=0, device="cpu", bias=False, group=GROUP) -> HQQWeights:
    """Synthetic layer of a given shape: random integers, scales ~2.7e-3*U(0.5,1.5),
    fractional zeros U(0, 2^b - 1) (values do not affect speed)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = n * k // group
    q = torch.randint(0, 2 ** nbits, (r, group), generator=g, dtype=torch.int32)
    scale = ((torch.rand(r, 1, generator=g) + 0.5) * 2.7e-3 * (16.0 / 2 ** nbits)).to(torch.float16)
    zero = (torch.rand(r, 1, generator=g) * (2 ** nbits - 1)).to(torch.float16)
    b = (torch.randn(n, generator=g) * 0.1).to(torch.float16) if bias else None
    return HQQWeights(pack_rows(q, nbits), scale, zero, nbits, (n, k), group, b).to(device)


def from_hqq_layer(layer) -> HQQWeights:
    """Adapt the reference's HQQLinear (or anything exposing .W_q/.meta/.bias,
    quantize.py:387-470) without importing it."""
    meta = layer.meta
    if meta.get("axis", 1) != 1:
        raise ValueError("only axis=1 HQQ layers are supported (AMQ uses axis=1)")
    if meta.get("view_as_float", False):
        raise ValueError("view_as_float HQQ payloads are not supported")
    if meta["group_size"] not in (64, 32) and (meta["group_size"] < GROUP or meta["group_size"] % GROUP):
        raise ValueError(f"group size must be 32, 64 or a multiple of 128 (got {meta['group_size']})")
    nbits = int(meta["nbits"])
    if nbits not in (2, 3, 4):
        raise NotImplementedError("Only 2,3,4 bits are supported.")
    W_q = layer.W_q.data if hasattr(layer.W_q, "data") else layer.W_q
    # compute_dtype = bfloat16 layers keep scale / zero in bf16 and dequantize in bf16 (quantize.py:184-199, 516): kept as they are (the bf16 entry
    # points reproduce that arithmetic); any other dtype goes to fp16, what the reference's kernels run in (ft.py:62)
    md = torch.bfloat16 if meta["scale"].dtype == torch.bfloat16 and meta["zero"].dtype == torch.bfloat16 else torch.float16
    return HQQWeights(W_q, meta["scale"].to(md).reshape(-1, 1), meta["zero"].to(md).reshape(-1, 1),
                      nbits, tuple(meta["shape"]), int(meta["group_size"]), getattr(layer, "bias", None), getattr(layer, "name", None))
