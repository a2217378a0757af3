#!/usr/bin/env python3
"""Golden vectors of the LayerNorm SITE of a quantized block at BERT-base width, made by RUNNING THE
REFERENCE's own classes -- model/util_layernorm.py:6-52 (QuantizedLayerNorm, QuantizedSplitLayerNorm, GammaResidual) on
top of its quantization package -- on one calibration batch [32,128,768]:

    observer pass   (observer on, fake-quant off):  y = wrapper(residual(x, hidden), lengths)   -> LayerNorm output, scale, zero_point
    quantized pass  (observer off, fake-quant on):  y = wrapper(residual(x, hidden), lengths)   -> the integer tensor x_quant

Stored per case: the un-quantised LayerNorm output of the first 2 samples (fp32), the integer tensor of ALL samples
(int8), scale / zero_point / observer statistics, the split bias, checksums of the re-drawn inputs.  The inputs are
not stored (tests/_ln_site.py re-draws them).  Outputs are DATA ONLY.  Runs in the build container (needs /root/reference):
    python tests/golden/make_golden_ln_site.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("OSQ_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(OUT))
from _ln_site import CASES, FLOAT_SAMPLES, SHAPE, checksum, ln_site_inputs  # noqa: E402  (tests/_ln_site.py)


class Cfg(dict):
    __getattr__ = dict.__getitem__


def main():
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from quant_transformer.model import util_layernorm as UL
    torch.set_num_threads(1)
    out = {}
    for name, cls, eps, with_gamma, quantizer, observer, pct, seed in CASES:
        x, hidden, gamma, beta, L = ln_site_inputs(seed)
        ln = torch.nn.LayerNorm(SHAPE[-1], eps=eps)
        with torch.no_grad():
            ln.weight.copy_(gamma)
            ln.bias.copy_(beta)
        cfg = Cfg(quantizer=quantizer, observer=observer, bit=6, symmetric=False, ch_axis=-1)
        mod = getattr(UL, cls)(ln, cfg, cfg, qoutput=True).eval()
        q = mod.layernorm_post_act_fake_quantize
        q.observer.set_name("encoder.layer.0.output.LayerNorm.layernorm_post_act_fake_quantize.observer")
        if pct is not None:
            q.observer.set_percentile(pct)
        res = UL.GammaResidual()
        if with_gamma:
            res.set_gamma(ln.weight.data)

        def site():
            r = x.clone() if with_gamma is None else res(x, hidden)
            return mod(r, L)
        with torch.no_grad():
            q.enable_observer()
            q.disable_fake_quant()
            y_obs = site()
            q.disable_observer()
            q.enable_fake_quant()
            y_q = site()
        scale, zp = q.scale.detach().reshape(-1), q.zero_point.detach().reshape(-1).float()
        xq = torch.round(y_q / scale + torch.round(zp))
        assert float(xq.min()) >= q.quant_min and float(xq.max()) <= q.quant_max
        # the integer tensor reproduces the reference's dequantised output bit for bit (util_quant.py:15: (x_q - zp) * scale)
        assert torch.equal((xq - torch.round(zp)) * scale, y_q)
        out[name + "_ln"] = y_obs[:FLOAT_SAMPLES].numpy()
        out[name + "_xq"] = xq.numpy().astype(np.int8)
        out[name + "_scale"], out[name + "_zp"] = scale.numpy(), zp.numpy()
        out[name + "_min"], out[name + "_max"] = q.observer.min_val.numpy(), q.observer.max_val.numpy()
        if hasattr(mod, "bias"):
            out[name + "_split_bias"] = mod.bias.data.numpy()
        out[name + "_sums"] = np.array([checksum(x), checksum(hidden), checksum(gamma), checksum(beta), int(L.sum())], dtype=np.int64)
        print(name, "scale", float(scale), "zp", float(zp), "valid tokens", int(L.sum()))
    path = os.path.join(OUT, "ln_site.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
