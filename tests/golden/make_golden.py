#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  It imports
the reference's ``quant_transformer.quantization`` package unmodified, with two
process-local shims that do not touch the reference files:

  * ``seaborn`` is imported but never used by observer.py:6 -> empty stub module;
  * ``Tensor.cuda()`` / ``Module.cuda()`` are hard-coded in observer.py:81,95,425
    and gamma_migration.py:67 -> identity (the reference CPU path is the parity
    target, see BASELINE.json ``north_star``).

Outputs are DATA ONLY (seeded inputs + what the reference returned), stored as
compressed .npz.  No reference source travels.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("OSQ_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from quant_transformer.quantization import observer, fake_quant, util_quant, quantized_module, state
    return observer, fake_quant, util_quant, quantized_module, state


O, FQ, U, QM, ST = _import_reference()
torch.set_num_threads(1)


class Cfg(dict):
    __getattr__ = dict.__getitem__


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


def activation_like(gen, shape, outlier_dims=2, outlier_gain=20.0):
    x = torch.randn(*shape, generator=gen)
    h = shape[-1]
    idx = torch.randperm(h, generator=gen)[:outlier_dims]
    x[..., idx] *= outlier_gain
    return x


# ---------------------------------------------------------------------------
# K1/K2: fake_quantize_per_tensor_affine / per_channel_affine  (util_quant.py:11-26)
# ---------------------------------------------------------------------------
def gen_fake_quant():
    gen = torch.Generator().manual_seed(1234)
    out = {}
    cases = []
    specials = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"),
                             1e-40, -1e-40, 3.4e38, -3.4e38, 1e-8, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5])
    k = 0
    for bit in (4, 6, 8):
        for sym in (False, True):
            ob = O.ObserverBase(bit=bit, symmetric=sym)
            qmin, qmax = ob.quant_min, ob.quant_max
            x = activation_like(gen, (4, 16, 32))
            mn, mx = x.min(), x.max()
            scale, zp = ob.calculate_qparams(mn, mx)
            scale_f, zp_i = scale.item(), int(zp.item())
            # values sitting exactly on .5 rounding boundaries and on the clamp edges
            ties = (torch.arange(qmin - 3, qmax + 4, dtype=torch.float32) - zp_i + 0.5) * scale_f
            xx = torch.cat([x.flatten(), ties, specials, specials * scale_f])
            y = U.fake_quantize_per_tensor_affine(xx, scale_f, zp_i, qmin, qmax)
            xq = torch.clamp(U.round_ste(xx / scale_f) + zp_i, qmin, qmax)
            out[f"pt{k}_x"], out[f"pt{k}_y"], out[f"pt{k}_xq"] = xx.numpy(), y.numpy(), xq.numpy()
            out[f"pt{k}_meta"] = np.array([scale_f, zp_i, qmin, qmax, bit, int(sym)], dtype=np.float64)
            cases.append(k)
            k += 1
    # tiny / floor scale and a large scale
    for scale_f, zp_i, qmin, qmax in ((1e-8, 0, -32, 31), (1.1920929e-07, 17, 0, 63), (1234.5, 100, 0, 255)):
        xx = torch.cat([torch.randn(500, generator=gen) * scale_f * 40, specials])
        y = U.fake_quantize_per_tensor_affine(xx, scale_f, zp_i, qmin, qmax)
        xq = torch.clamp(U.round_ste(xx / scale_f) + zp_i, qmin, qmax)
        out[f"pt{k}_x"], out[f"pt{k}_y"], out[f"pt{k}_xq"] = xx.numpy(), y.numpy(), xq.numpy()
        out[f"pt{k}_meta"] = np.array([scale_f, zp_i, qmin, qmax, 0, 0], dtype=np.float64)
        k += 1
    out["n_per_tensor"] = k

    # per-channel: weights [C_out, C_in] ch_axis=0 (sym, 4/6 bit) and an activation ch_axis=2 (asym)
    c = 0
    for (shape, ch_axis, bit, sym) in (((24, 40), 0, 6, True), ((7, 33), 0, 4, True), ((3, 5, 16), 2, 6, False),
                                       ((6, 4, 3, 3), 0, 8, True), ((4, 6, 10), 1, 6, False)):
        ob = O.MinMaxObserver(bit=bit, symmetric=sym, ch_axis=ch_axis)
        w = torch.randn(*shape, generator=gen) * 0.05
        ob(w)
        scale, zp = ob.calculate_qparams(ob.min_val, ob.max_val)
        y = U.fake_quantize_per_channel_affine(w, scale, zp.int(), ch_axis, ob.quant_min, ob.quant_max)
        ns = [1] * w.dim()
        ns[ch_axis] = w.shape[ch_axis]
        xq = torch.clamp(U.round_ste(w / scale.reshape(ns)) + zp.int().reshape(ns), ob.quant_min, ob.quant_max)
        out[f"pc{c}_x"], out[f"pc{c}_y"], out[f"pc{c}_xq"] = w.numpy(), y.numpy(), xq.numpy()
        out[f"pc{c}_scale"], out[f"pc{c}_zp"] = scale.numpy(), zp.int().numpy()
        out[f"pc{c}_min"], out[f"pc{c}_max"] = ob.min_val.numpy(), ob.max_val.numpy()
        out[f"pc{c}_meta"] = np.array([ch_axis, ob.quant_min, ob.quant_max, bit, int(sym)], dtype=np.int64)
        c += 1
    out["n_per_channel"] = c
    save("fake_quant", **out)


# ---------------------------------------------------------------------------
# K3: LSQ+ forward and backward via autograd (util_quant.py:48-67)
# ---------------------------------------------------------------------------
def gen_lsqplus():
    gen = torch.Generator().manual_seed(4321)
    out = {}
    k = 0
    for (shape, bit, sym, zp0) in (((4, 16, 32), 6, False, 23.3), ((2, 3, 8, 16), 8, False, 127.5),
                                   ((64, 48), 4, False, 7.49), ((8, 8, 24), 6, True, 0.0)):
        qmin, qmax = O.ObserverBase(bit=bit, symmetric=sym).quant_min, O.ObserverBase(bit=bit, symmetric=sym).quant_max
        x = activation_like(gen, shape).requires_grad_(True)
        span = (x.max() - x.min()).item()
        scale = torch.tensor([span / (qmax - qmin) * 0.7], requires_grad=True)
        zp = torch.tensor([zp0], requires_grad=True)
        g = 1.0 / (x.numel() * qmax) ** 0.5
        y = U.fake_quantize_learnableplus_per_tensor_affine_training(x, scale, zp, qmin, qmax, g)
        gy = torch.randn(*shape, generator=gen)
        y.backward(gy)
        out[f"c{k}_x"], out[f"c{k}_gy"], out[f"c{k}_y"] = x.detach().numpy(), gy.numpy(), y.detach().numpy()
        out[f"c{k}_dx"], out[f"c{k}_ds"], out[f"c{k}_dzp"] = x.grad.numpy(), scale.grad.numpy(), zp.grad.numpy()
        out[f"c{k}_meta"] = np.array([scale.item(), zp.item(), qmin, qmax, g], dtype=np.float64)
        k += 1
    out["n"] = k
    # per-channel LSQ+ forward/backward (ch_axis = 0)
    x = (torch.randn(12, 20, generator=gen) * 0.1).requires_grad_(True)
    scale = (torch.rand(12, generator=gen) * 0.01 + 0.002).requires_grad_(True)
    zp = (torch.rand(12, generator=gen) * 10 + 20).requires_grad_(True)
    g = 1.0 / (x.numel() / 12 * 63) ** 0.5
    y = U.fake_quantize_learnableplus_per_channel_affine_training(x, scale, zp, 0, 0, 63, g)
    gy = torch.randn(12, 20, generator=gen)
    y.backward(gy)
    out["pc_x"], out["pc_gy"], out["pc_y"] = x.detach().numpy(), gy.numpy(), y.detach().numpy()
    out["pc_scale"], out["pc_zp"] = scale.detach().numpy(), zp.detach().numpy()
    out["pc_dx"], out["pc_ds"], out["pc_dzp"] = x.grad.numpy(), scale.grad.numpy(), zp.grad.numpy()
    out["pc_meta"] = np.array([0, 0, 63, g], dtype=np.float64)
    save("lsqplus", **out)


# ---------------------------------------------------------------------------
# K9: calculate_qparams (observer.py:101-119)
# ---------------------------------------------------------------------------
def gen_qparams():
    gen = torch.Generator().manual_seed(99)
    out = {}
    mins = torch.cat([torch.randn(200, generator=gen) * 3 - 1, torch.tensor([0.0, 1.0, -1e-12, 5.0, -7.0, 0.0, -3.0])])
    maxs = torch.cat([mins[:200] + torch.rand(200, generator=gen) * 6, torch.tensor([0.0, 2.0, 1e-12, 9.0, -2.0, 4.0, 0.0])])
    k = 0
    for bit in (4, 6, 8):
        for sym in (False, True):
            ob = O.ObserverBase(bit=bit, symmetric=sym)
            scale, zp = ob.calculate_qparams(mins, maxs)
            out[f"c{k}_scale"], out[f"c{k}_zp"] = scale.numpy(), zp.numpy()
            out[f"c{k}_meta"] = np.array([bit, int(sym), ob.quant_min, ob.quant_max], dtype=np.int64)
            k += 1
    out["min"], out["max"], out["n"] = mins.numpy(), maxs.numpy(), k
    save("qparams", **out)


# ---------------------------------------------------------------------------
# K4-K8: observer sequences (observer.py:122-237)
# ---------------------------------------------------------------------------
def gen_observers():
    gen = torch.Generator().manual_seed(2024)
    out = {}
    B, T, H, h = 4, 16, 32, 2
    d = H // h
    layouts = {
        "bth": (lambda: activation_like(gen, (B, T, H)), 1),
        "bhtd": (lambda: activation_like(gen, (B, T, H)).view(B, T, h, d).permute(0, 2, 1, 3), 2),
        "bhdt": (lambda: activation_like(gen, (B, T, H)).view(B, T, h, d).permute(0, 2, 1, 3).transpose(-1, -2), 3),
        "bhtt": (lambda: torch.softmax(torch.randn(B, h, T, T, generator=gen) * 3, dim=-1), 2),
        "bart3d": (lambda: torch.softmax(torch.randn(B * h, T, T, generator=gen) * 3, dim=-1), 1),
        "bhtd_contig": (lambda: activation_like(gen, (B, h, T, d)), 2),
    }
    percentiles = [1.0, 0.99, 0.97, 0.9, 0.71]
    k = 0
    for obs_name in ("MinMaxObserver", "AvgMinMaxObserver", "AvgPruneMinMaxObserver"):
        for lay, (make, seq_pos) in layouts.items():
            for masked in (True, False):
                if obs_name != "AvgPruneMinMaxObserver" and not masked and lay != "bth":
                    continue
                names = ["encoder.layer.0.x_post_act_fake_quantize.observer"]
                if obs_name == "AvgPruneMinMaxObserver" and lay in ("bhtt", "bart3d"):
                    names.append("encoder.layer.0.attention_probs_post_act_fake_quantize.observer")
                for name in names:
                    plist = percentiles if obs_name == "AvgPruneMinMaxObserver" and "attention_probs" not in name else [None]
                    for p in plist:
                        ob = QM.ObserverDict[obs_name](bit=6, symmetric=False, ch_axis=-1)
                        ob.set_name(name)
                        if p is not None:
                            ob.set_percentile(p)
                        xs, lens, mins, maxs = [], [], [], []
                        for it in range(3):
                            x = make()
                            L = torch.randint(1, T + 1, (B,), generator=gen)
                            if it == 1:
                                L[0] = T  # one full-length sample
                            ob(x, observation_mask=L if masked else None, seq_pos=seq_pos if (masked or obs_name == "AvgPruneMinMaxObserver") else -1)
                            xs.append(x.contiguous().numpy().copy())
                            lens.append(L.numpy().copy())
                            mins.append(ob.min_val.numpy().copy())
                            maxs.append(ob.max_val.numpy().copy())
                        scale, zp = ob.calculate_qparams(ob.min_val, ob.max_val)
                        out[f"c{k}_x"], out[f"c{k}_len"] = np.stack(xs), np.stack(lens)
                        out[f"c{k}_min"], out[f"c{k}_max"] = np.stack(mins), np.stack(maxs)
                        out[f"c{k}_scale"], out[f"c{k}_zp"] = scale.numpy(), zp.numpy()
                        out[f"c{k}_info"] = np.array([obs_name, lay, str(seq_pos), str(int(masked)), name,
                                                      "" if p is None else repr(p)])
                        k += 1
    # [B, H] pooler-style call: no mask, seq_pos=-1 -> no pruning even for AvgPrune (observer.py:220-226)
    for obs_name in ("MinMaxObserver", "AvgMinMaxObserver", "AvgPruneMinMaxObserver"):
        ob = QM.ObserverDict[obs_name](bit=6, symmetric=False, ch_axis=-1)
        ob.set_name("pooler.getitem_post_act_fake_quantize.observer")
        ob.set_percentile(0.9)
        xs, mins, maxs = [], [], []
        for it in range(3):
            x = activation_like(gen, (B, H))
            ob(x)
            xs.append(x.numpy().copy()); mins.append(ob.min_val.numpy().copy()); maxs.append(ob.max_val.numpy().copy())
        out[f"c{k}_x"], out[f"c{k}_len"] = np.stack(xs), np.zeros((3, 0), dtype=np.int64)
        out[f"c{k}_min"], out[f"c{k}_max"] = np.stack(mins), np.stack(maxs)
        scale, zp = ob.calculate_qparams(ob.min_val, ob.max_val)
        out[f"c{k}_scale"], out[f"c{k}_zp"] = scale.numpy(), zp.numpy()
        out[f"c{k}_info"] = np.array([obs_name, "bh", "-1", "0", ob.name, "0.9"])
        k += 1
    out["n"] = k
    save("observers", **out)

    # one mid-size case (B=8, T=64, H=256): per-token extrema and pruned range only (inputs regenerated from seed)
    gen2 = torch.Generator().manual_seed(77)
    x = activation_like(gen2, (8, 64, 256), outlier_dims=3)
    L = torch.randint(4, 65, (8,), generator=gen2)
    res = {}
    for p in (1.0, 0.99, 0.95, 0.9, 0.8, 0.71):
        ob = O.AvgPruneMinMaxObserver(bit=6, symmetric=False)
        ob.set_name("layer.x_post_act_fake_quantize.observer"); ob.set_percentile(p)
        ob(x, observation_mask=L, seq_pos=1)
        res[repr(p)] = (ob.min_val.item(), ob.max_val.item())
    v = ob.remove_padding(x, L, 1)
    save("observer_midsize", lengths=L.numpy(), token_max=v.max(1)[0].numpy(), token_min=v.min(1)[0].numpy(),
         percentiles=np.array([float(p) for p in res]), mins=np.array([m for m, _ in res.values()], dtype=np.float32),
         maxs=np.array([m for _, m in res.values()], dtype=np.float32), seed=np.array([77]))


# ---------------------------------------------------------------------------
# K10: MSEFast / AvgMSEFast (observer.py:412-567)
# ---------------------------------------------------------------------------
def gen_msefast():
    gen = torch.Generator().manual_seed(555)
    out = {}
    k = 0

    def run(cls, x, bit, sym, ch_axis, reps=1):
        ob = cls(bit=bit, symmetric=sym, ch_axis=ch_axis)
        nfev = [0]
        orig = ob.loss_fx

        def counted(*a, **kw):
            nfev[0] += 1
            return orig(*a, **kw)
        ob.loss_fx = counted
        mins, maxs = [], []
        for r in range(reps):
            ob(x[r] if reps > 1 else x)
            mins.append(np.asarray(ob.min_val.numpy()).copy()); maxs.append(np.asarray(ob.max_val.numpy()).copy())
        return np.stack(mins), np.stack(maxs), nfev[0], ob.one_side_dist

    specs = [
        ("MSEFastObserver", activation_like(gen, (4, 16, 32)), 6, True, -1, 1),          # 1-D symmetric
        ("MSEFastObserver", torch.relu(activation_like(gen, (4, 16, 32))), 6, False, -1, 1),  # one-sided 'pos'
        ("MSEFastObserver", -torch.relu(activation_like(gen, (4, 16, 32))), 6, False, -1, 1),  # one-sided 'neg'
        ("MSEFastObserver", activation_like(gen, (4, 16, 32)), 6, False, -1, 1),          # 2-D
        ("MSEFastObserver", torch.randn(16, 48, generator=gen) * 0.05, 4, True, 0, 1),    # per-channel weights
        ("AvgMSEFastObserver", torch.stack([activation_like(gen, (2, 8, 32)) for _ in range(3)]), 6, False, -1, 3),
        ("AvgMSEFastObserver", torch.stack([activation_like(gen, (2, 8, 32)) for _ in range(3)]), 4, True, -1, 3),
    ]
    for (cls_name, x, bit, sym, ch_axis, reps) in specs:
        mins, maxs, nfev, osd = run(QM.ObserverDict[cls_name], x, bit, sym, ch_axis, reps)
        out[f"c{k}_x"], out[f"c{k}_min"], out[f"c{k}_max"] = x.numpy(), mins, maxs
        out[f"c{k}_info"] = np.array([cls_name, str(bit), str(int(sym)), str(ch_axis), str(reps), str(nfev), osd])
        k += 1
    out["n"] = k
    save("msefast", **out)


def gen_msefast_masked():
    """Per-tensor MSEFast / AvgMSEFast on MASKED activations (observation_mask + seq_pos: remove_padding lays the valid
    tokens out sample after sample, observer.py:72-84), three batches each, so that the second and third call run on a
    float64 copy of x (observer.py:524,549): statistics after every call and the total number of loss evaluations.
    Same torch.set_num_threads(1) as everything here: the sums inside the loss run ATen's serial cascade."""
    gen = torch.Generator().manual_seed(777)
    out = {}
    k = 0
    specs = [("AvgMSEFastObserver", (4, 16, 32), 6, False, 1), ("AvgMSEFastObserver", (4, 16, 32), 6, True, 1),
             ("MSEFastObserver", (3, 12, 48), 4, False, 1), ("AvgMSEFastObserver", (2, 4, 10, 16), 6, False, 2)]
    for cls_name, shape, bit, sym, seq_pos in specs:
        ob = QM.ObserverDict[cls_name](bit=bit, symmetric=sym, ch_axis=-1)
        nfev = [0]
        orig = ob.loss_fx

        def counted(*a, _orig=orig, **kw):
            nfev[0] += 1
            return _orig(*a, **kw)
        ob.loss_fx = counted
        xs, lens, mins, maxs = [], [], [], []
        for r in range(3):
            x = activation_like(gen, shape) * (1.0 + 0.4 * r)
            T = shape[seq_pos]
            L = torch.randint(1, T + 1, (shape[0],), generator=gen)
            L[int(torch.randint(0, shape[0], (1,), generator=gen))] = T
            ob(x, L, seq_pos)
            xs.append(x.numpy()); lens.append(L.numpy())
            mins.append(np.asarray(ob.min_val.numpy()).copy()); maxs.append(np.asarray(ob.max_val.numpy()).copy())
        out[f"c{k}_x"], out[f"c{k}_len"] = np.stack(xs), np.stack(lens)
        out[f"c{k}_min"], out[f"c{k}_max"] = np.stack(mins), np.stack(maxs)
        out[f"c{k}_info"] = np.array([cls_name, str(bit), str(int(sym)), str(seq_pos), str(nfev[0]), ob.one_side_dist])
        k += 1
    out["n"] = k
    save("msefast_masked", **out)


MSEFAST_ROW_CASES = (("w768", 901, 2048, 768, 4), ("w3072", 902, 2048, 3072, 4), ("w768_6bit", 903, 1024, 768, 6))


def msefast_row_weights(seed, rows, cols):
    """The seeded weight matrix of a MSEFAST_ROW_CASES entry (the tests rebuild it from the seed: only the
    reference's per-row results are stored)."""
    return torch.randn(rows, cols, generator=torch.Generator().manual_seed(seed)) * 0.05


def gen_msefast_rows():
    """BERT-base row sizes (768 / 3072 columns), thousands of rows, per-channel symmetric MSEFastObserver: the
    reference's own per-row (min, max) for the deviation statistics of tests/test_gpu_parity.py
    ::test_msefast_rows_against_reference and tests/test_oracle_golden.py (observer.py:496-517)."""
    out = {}
    for name, seed, rows, cols, bit in MSEFAST_ROW_CASES:
        w = msefast_row_weights(seed, rows, cols)
        ob = O.MSEFastObserver(bit=bit, symmetric=True, ch_axis=0)
        nfev = [0]
        orig = ob.loss_fx

        def counted(*a, _orig=orig, **kw):
            nfev[0] += 1
            return _orig(*a, **kw)
        ob.loss_fx = counted
        ob(w)
        out[name + "_min"], out[name + "_max"] = ob.min_val.numpy().copy(), ob.max_val.numpy().copy()
        out[name + "_info"] = np.array([seed, rows, cols, bit, nfev[0]])
        print(name, "nfev", nfev[0])
    save("msefast_rows", **out)


# ---------------------------------------------------------------------------
# A7/A8/A17/A18: module-level traces through Quantizer / state togglers
# ---------------------------------------------------------------------------
def gen_modules():
    gen = torch.Generator().manual_seed(31337)
    out = {}
    k = 0
    B, T, H = 4, 16, 32
    for (quantizer, observer, bit, sym, ch_axis, kind) in (
            ("FixedFakeQuantize", "AvgMinMaxObserver", 6, False, -1, "act"),
            ("FixedFakeQuantize", "MinMaxObserver", 6, True, 0, "weight"),
            ("LSQPlusFakeQuantize", "AvgPruneMinMaxObserver", 6, False, -1, "act"),
            ("LSQFakeQuantize", "MinMaxObserver", 8, True, -1, "act"),
            ("FixedFakeQuantize", "MinMaxObserver", 4, False, -1, "act"),
            ("LSQPlusFakeQuantize", "MinMaxObserver", 6, False, 0, "weight")):
        cfg = Cfg(quantizer=quantizer, observer=observer, bit=bit, symmetric=sym, ch_axis=ch_axis)
        q = QM.Quantizer(None, cfg)
        q.observer.set_name("m.x_post_act_fake_quantize.observer")
        q.observer.set_percentile(0.9)
        q.enable_observer(); q.disable_fake_quant()
        xs, lens, scales, zps = [], [], [], []
        for it in range(3 if kind == "act" else 1):
            x = activation_like(gen, (B, T, H)) if kind == "act" else torch.randn(24, 40, generator=gen) * 0.05
            L = torch.randint(1, T + 1, (B,), generator=gen)
            r = q(x, observation_mask=L, seq_pos=1) if kind == "act" else q(x)
            assert r is x
            xs.append(x.numpy().copy()); lens.append(L.numpy().copy())
            scales.append(q.scale.detach().numpy().copy()); zps.append(q.zero_point.detach().numpy().copy())
        q.disable_observer(); q.enable_fake_quant()
        xin = xs[-1]
        xt = torch.from_numpy(xin).clone().requires_grad_(True)
        y = q(xt, observation_mask=torch.from_numpy(lens[-1]), seq_pos=1) if kind == "act" else q(xt)
        gy = torch.randn(*xin.shape, generator=gen)
        y.backward(gy)
        out[f"c{k}_x"], out[f"c{k}_len"] = np.stack(xs), np.stack(lens)
        out[f"c{k}_scale"], out[f"c{k}_zp"] = np.stack(scales), np.stack(zps)
        out[f"c{k}_y"], out[f"c{k}_gy"], out[f"c{k}_dx"] = y.detach().numpy(), gy.numpy(), xt.grad.numpy()
        if isinstance(q.scale, torch.nn.Parameter) and q.scale.grad is not None:
            out[f"c{k}_ds"] = q.scale.grad.numpy()
        if isinstance(q.zero_point, torch.nn.Parameter) and q.zero_point.grad is not None:
            out[f"c{k}_dzp"] = q.zero_point.grad.numpy()
        sd = q.state_dict()
        out[f"c{k}_sdkeys"] = np.array(sorted(sd.keys()))
        out[f"c{k}_info"] = np.array([quantizer, observer, str(bit), str(int(sym)), str(ch_axis), kind,
                                      str(q.scale.dtype), str(q.zero_point.dtype)])
        k += 1
    out["n"] = k
    save("modules", **out)


# ---------------------------------------------------------------------------
# K11/K12: gamma migration pieces (gamma_migration.py:63-71, util_layernorm.py:21-52)
# ---------------------------------------------------------------------------
def gen_gamma():
    sys.path.insert(0, REF)
    from quant_transformer.model.util_layernorm import QuantizedSplitLayerNorm, GammaResidual, QuantizedLayerNorm
    gen = torch.Generator().manual_seed(808)
    Hn = 48
    ln = torch.nn.LayerNorm(Hn, eps=1e-12)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(Hn, generator=gen) * 1.5 + 0.2)
        ln.weight[3] = 6.0  # an "outlier" gamma
        ln.bias.copy_(torch.randn(Hn, generator=gen) * 0.3)
    cfg = Cfg(quantizer="FixedFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    x = activation_like(gen, (3, 10, Hn))
    hidden = torch.randn(3, 10, Hn, generator=gen)
    qln = QuantizedLayerNorm(ln, cfg, cfg, qoutput=True)
    split = QuantizedSplitLayerNorm(ln, cfg, cfg, qoutput=True).eval()
    res = GammaResidual()
    y_before = res(x, hidden)
    res.set_gamma(ln.weight.data)
    y_after = res(x, hidden)
    with torch.no_grad():
        ln_full = qln(x.clone())
        ln_split = split(x.clone())
    W = torch.randn(20, Hn, generator=gen) * 0.05
    Wf = W.clone()
    Wf *= ln.weight.data.detach().clone()
    save("gamma", x=x.numpy(), hidden=hidden.numpy(), gamma=ln.weight.data.numpy(), beta=ln.bias.data.numpy(),
         ln_full=ln_full.numpy(), ln_split=ln_split.numpy(), split_bias=split.bias.data.numpy(),
         res_before=y_before.detach().numpy(), res_after=y_after.detach().numpy(), W=W.numpy(), W_folded=Wf.numpy())


# ---------------------------------------------------------------------------
# remaining observers of ObserverDict: LSQPlusObserver, AvgQuantileObserver, MSEObserver, AvgMSEObserver
# (observer.py:148-173, 240-282, 285-409)
# ---------------------------------------------------------------------------
def gen_other_observers():
    gen = torch.Generator().manual_seed(4242)
    out = {}
    # LSQPlusObserver: mean +- 3 std, symmetric only
    x = activation_like(gen, (4, 16, 32))
    ob = O.LSQPlusObserver(bit=8, symmetric=True, ch_axis=-1)
    ob(x)
    out["lsqp_x"], out["lsqp_min"], out["lsqp_max"] = x.numpy(), ob.min_val.numpy(), ob.max_val.numpy()
    w = torch.randn(24, 40, generator=gen) * 0.05
    ob = O.LSQPlusObserver(bit=4, symmetric=True, ch_axis=0)
    ob(w)
    s, z = ob.calculate_qparams(ob.min_val, ob.max_val)
    out["lsqp_w"], out["lsqp_wmin"], out["lsqp_wmax"], out["lsqp_wscale"] = w.numpy(), ob.min_val.numpy(), ob.max_val.numpy(), s.numpy()
    # AvgQuantileObserver: 2048-bin histogram of |x|, clip where the cumulative count reaches threshold*numel
    k = 0
    for threshold, masked in ((0.99999, True), (0.99, True), (0.9, False), (0.999, True)):
        ob = O.AvgQuantileObserver(bit=6, symmetric=False, ch_axis=-1, threshold=threshold)
        xs, lens, mins, maxs = [], [], [], []
        for it in range(3):
            x = activation_like(gen, (4, 16, 32))
            L = torch.randint(1, 17, (4,), generator=gen)
            ob(x, observation_mask=L if masked else None, seq_pos=1 if masked else -1)
            xs.append(x.numpy().copy()); lens.append(L.numpy().copy())
            mins.append(np.asarray(ob.min_val.numpy()).copy()); maxs.append(np.asarray(ob.max_val.numpy()).copy())
        out[f"aq{k}_x"], out[f"aq{k}_len"] = np.stack(xs), np.stack(lens)
        out[f"aq{k}_min"], out[f"aq{k}_max"] = np.stack(mins), np.stack(maxs)
        out[f"aq{k}_meta"] = np.array([threshold, float(masked)])
        k += 1
    out["aq_n"] = k
    # one larger histogram case to pin the binning itself
    xl = activation_like(gen, (8, 64, 96), outlier_dims=3)
    mx = torch.max(-xl.min(), xl.max())
    out["hist_x"], out["hist_counts"] = xl.numpy(), torch.histc(torch.abs(xl), bins=2048, min=0.0, max=mx).numpy()
    # MSEObserver / AvgMSEObserver: brute-force grid (100 ranges x (qmax-qmin+1) zero-points in 2-D)
    k = 0
    specs = [("MSEObserver", activation_like(gen, (4, 16, 32)), 4, True, -1, 1),                      # 1-D symmetric
             ("MSEObserver", torch.relu(activation_like(gen, (4, 16, 32))), 4, False, -1, 1),          # 1-D one-sided
             ("MSEObserver", activation_like(gen, (4, 16, 32)), 4, False, -1, 1),                      # 2-D (100 x 16)
             ("MSEObserver", torch.randn(12, 40, generator=gen) * 0.05, 4, True, 0, 1),               # per-channel 1-D
             ("AvgMSEObserver", torch.stack([activation_like(gen, (2, 8, 32)) for _ in range(2)]), 4, False, -1, 2)]
    for (cls_name, x, bit, sym, ch_axis, reps) in specs:
        ob = QM.ObserverDict[cls_name](bit=bit, symmetric=sym, ch_axis=ch_axis)
        mins, maxs = [], []
        for r in range(reps):
            ob(x[r] if reps > 1 else x)
            mins.append(np.asarray(ob.min_val.numpy()).copy()); maxs.append(np.asarray(ob.max_val.numpy()).copy())
        out[f"mse{k}_x"], out[f"mse{k}_min"], out[f"mse{k}_max"] = x.numpy(), np.stack(mins), np.stack(maxs)
        out[f"mse{k}_info"] = np.array([cls_name, str(bit), str(int(sym)), str(ch_axis), str(reps), ob.one_side_dist])
        k += 1
    out["mse_n"] = k
    save("other_observers", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "other":
        gen_other_observers()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "msefast_masked":
        gen_msefast_masked()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "msefast_rows":
        gen_msefast_rows()
        sys.exit(0)
    gen_fake_quant()
    gen_lsqplus()
    gen_qparams()
    gen_observers()
    gen_msefast()
    gen_msefast_masked()
    gen_msefast_rows()
    gen_modules()
    gen_gamma()
    gen_other_observers()
