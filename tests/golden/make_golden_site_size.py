#!/usr/bin/env python3
"""Golden vectors at the SITE SIZES of BASELINE configs[3] (RoBERTa / BERT-base, batches of [32,128]), made by RUNNING THE
REFERENCE on one thread (torch.set_num_threads(1): ATen's serial cascade_sum then holds at any length, so the order in
which the reference adds its squared errors / gradient terms is defined and reproducible):

  * per-tensor AvgMSEFastObserver / MSEFastObserver (quantization/observer.py:412-567) on
      - [32,128,768]      masked hidden states, two-sided        -> nested 2-D search, float64 arithmetic from call 2 on
      - [32,12,128,128]   attention probabilities, non-negative  -> 1-D search, fp32 for ever (observer.py:491)
      - [32,128,3072]     GELU outputs, two-sided                -> nested 2-D search
    statistics after every call and the number of loss evaluations;
  * LSQ+ fake-quant forward + autograd backward (quantization/util_quant.py:48-55,70-71) on [32,128,768] and
    [32,128,3072]: scale.grad, zero_point.grad and a checksum of x.grad.

The inputs are NOT stored (12-50 MB each): they are drawn from seeded torch CPU generators, which the tests re-run
(`site_input`, same torch build in the image); the int64 sums of their bit patterns are stored to catch a generator that drifted.
Outputs are DATA ONLY.  Runs in the build container (needs /root/reference).  ~15 minutes on one core:
    python tests/golden/make_golden_site_size.py
"""
import os
import sys
import time
import types

import numpy as np
import torch

REF = os.environ.get("OSQ_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.path.insert(0, os.path.dirname(OUT))
from _site_size import BWD_CASES, MSE_CASES, bwd_case, checksum, site_input, site_lengths  # noqa: E402  (tests/_site_size.py)


def main():
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from quant_transformer.quantization import quantized_module as QM, util_quant as U
    torch.set_num_threads(1)
    out = {}
    for name, cls_name, shape, seq_pos, kind, bit, sym, batches, seed in MSE_CASES:
        t0 = time.time()
        gen = torch.Generator().manual_seed(seed)
        ob = QM.ObserverDict[cls_name](bit=bit, symmetric=sym, ch_axis=-1)
        nfev = [0]
        orig = ob.loss_fx

        def counted(*a, _orig=orig, _n=nfev, **kw):
            _n[0] += 1
            return _orig(*a, **kw)
        ob.loss_fx = counted
        mins, maxs, sums, evals = [], [], [], []
        for r in range(batches):
            x = site_input(gen, shape, kind, r)
            L = site_lengths(gen, shape, seq_pos)
            ob(x, L, seq_pos)
            mins.append(np.asarray(ob.min_val.numpy(), dtype=np.float64).copy())
            maxs.append(np.asarray(ob.max_val.numpy(), dtype=np.float64).copy())
            sums.append(checksum(x))
            evals.append(nfev[0])
        out[f"{name}_min"], out[f"{name}_max"] = np.stack(mins), np.stack(maxs)
        out[f"{name}_xsum"], out[f"{name}_nfev"] = np.array(sums), np.array(evals)
        out[f"{name}_side"] = np.array(ob.one_side_dist)
        print(name, "nfev", evals, "min", mins[-1], "max", maxs[-1], f"{time.time() - t0:.0f} s", flush=True)
    for name, shape, kind, seed in BWD_CASES:
        x, gy, scale, zp, g = bwd_case(shape, kind, seed)
        x.requires_grad_(True), scale.requires_grad_(True), zp.requires_grad_(True)
        y = U.fake_quantize_learnableplus_per_tensor_affine_training(x, scale, zp, 0, 63, g)
        y.backward(gy)
        out[f"{name}_scale"], out[f"{name}_zp"] = scale.detach().numpy(), zp.detach().numpy()
        out[f"{name}_dscale"], out[f"{name}_dzp"] = scale.grad.numpy(), zp.grad.numpy()
        out[f"{name}_xsum"] = np.array([checksum(x), checksum(gy)])
        out[f"{name}_dxsum"] = np.array([checksum(x.grad)])
        out[f"{name}_ysum"] = np.array([checksum(y)])
        print(name, "dscale", scale.grad.item(), "dzp", zp.grad.item(), flush=True)
    path = os.path.join(OUT, "site_size.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
