#!/usr/bin/env python3
"""Golden vectors for group sizes other than 128 (256; 64 and 32), produced by the REAL reference on CPU -- same recipe, stubs and rules as
gen_golden.py (data only; no reference source travels).  Separate script so that the group-128 fixtures are not regenerated.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_groups.py

Per case ``hqq_g{G}_b{bits}_{N}x{K}.npz``: W, W_q, scale, zero, W_deq, x[3,K], y_ref, and the GPTQLinear buffers after
patch_hqq_to_gptq (qweight int32, scales / zeros fp32 [K/G, N]) with GPTQLinear.forward on 128 rows (torch fallback branch).
(The 256 cases come first, from the seed they have always had: regenerating leaves them byte-identical.)"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import _stubs, REF, HERE      # noqa: E402


def main():
    sys.dont_write_bytecode = True
    sys.path[:0] = [_stubs(), REF]
    import torch
    from hqq.core.quantize import HQQLinear, BaseQuantizeConfig
    from hqq.backends.autogptq import patch_hqq_to_gptq

    for G, seed in ((256, 4321), (64, 6464), (32, 3232)):
      torch.manual_seed(seed)
      for bits, n, k in ((2, 64, 512), (3, 64, 512), (4, 64, 512)):
          lin = torch.nn.Linear(k, n, bias=False)
          with torch.no_grad():
              lin.weight.copy_(torch.randn(n, k) * 0.02)
              lin.weight[::5, ::11] *= 4.0
          lin = lin.half()
          cfg = BaseQuantizeConfig(nbits=bits, group_size=G, axis=1)

          def make():
              hh = HQQLinear(copy.deepcopy(lin), cfg, compute_dtype=torch.float16, device="cpu", del_orig=False)
              hh.name = "golden"
              return hh
          h = make()
          assert int(h.meta["group_size"]) == G
          x = torch.randn(3, k).half()
          out = {"W": lin.weight.data.numpy(), "W_q": h.W_q.numpy(), "scale": h.meta["scale"].numpy(), "zero": h.meta["zero"].numpy(),
                 "W_deq": h.dequantize().numpy(), "nbits": np.int32(bits), "group_size": np.int32(G), "shape": np.array([n, k], np.int32),
                 "x": x.numpy(), "y_ref": torch.matmul(x, h.dequantize().T).numpy()}
          g = patch_hqq_to_gptq(make(), None)
          out["gptq_qweight"], out["gptq_scales"], out["gptq_zeros"] = g.qweight.numpy(), g.scales.numpy(), g.zeros.numpy()
          gx = torch.randn(128, k).half()
          out["gptq_x"], out["gptq_y"] = gx.numpy(), g(gx).detach().numpy()
          np.savez_compressed(f"{HERE}/hqq_g{G}_b{bits}_{n}x{k}.npz", **out)
          print("wrote", f"hqq_g{G}_b{bits}_{n}x{k}.npz", {k_: v.shape for k_, v in out.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    main()
