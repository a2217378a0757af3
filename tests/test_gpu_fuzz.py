"""Randomised differential test: the module API on the GPU against the oracle over layouts, sizes and value patterns no
hand-written case lists -- odd shapes (T < 4, 1..7 features), permuted and offset (16-byte-misaligned) views, empty and
full samples, one-sided and constant tensors, heavy duplicates (quantised values), 4 / 6 / 8 bit, both symmetries, the
three running-statistic observers, Fixed and LSQ+ quantizers, forward and LSQ+ backward.  Seeded: the cases are the same
in every run; OSQ_FUZZ_CASES=<n> lengthens the walk, OSQ_FUZZ_SEED=<n> takes another one (default 800 cases, a few seconds)."""
import os
from types import SimpleNamespace as NS

import numpy as np
from conftest import bits_equal
import pytest
import torch

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("OSQ_FUZZ_CASES", "800"))
SEED = int(os.environ.get("OSQ_FUZZ_SEED", "0"))          # another walk: OSQ_FUZZ_SEED=<n> shifts every family's seed


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd import _hip
    _hip.load()
    return torch.device("cuda:0")


def _draw_shape(rng):
    kind = rng.choice(["bth", "bhtd", "bhdt", "bh", "probs", "probs3d"])
    small = rng.random() < 0.5
    B = int(rng.integers(1, 6 if small else 12))
    T = int(rng.integers(1, 9 if small else 70))
    if kind == "bth":
        H = int(rng.choice([1, 2, 3, 5, 7, 8, 12, 64, 100, 256, 260, 768]))
        return kind, (B, T, H), 1
    if kind == "bh":
        return kind, (B, int(rng.choice([1, 3, 8, 33, 768]))), -1
    if kind == "probs3d":          # BART: [B*h, T, S] probabilities with a length-B mask (remove_padding's zip stops after B rows)
        return kind, (B * int(rng.integers(1, 5)), T, int(rng.integers(1, 40))), 1
    h = int(rng.integers(1, 5))
    d = int(rng.choice([1, 2, 4, 6, 8, 16, 64]))
    if kind == "bhtd":
        return kind, (B, h, T, d), 2
    if kind == "bhdt":
        return kind, (B, h, d, T), 3
    return kind, (B, h, T, T), 2


def _draw_values(rng, shape, pattern):
    x = rng.standard_normal(shape).astype(np.float32)
    if pattern == "outlier":
        x[..., 0] *= 25.0
    elif pattern == "positive":
        x = np.abs(x) + np.float32(0.01)
    elif pattern == "negative":
        x = -np.abs(x) - np.float32(0.01)
    elif pattern == "constant":
        x[...] = np.float32(rng.choice([0.0, 1.5, -2.25]))
    elif pattern == "duplicates":
        x = np.round(x * 2.0).astype(np.float32) / np.float32(2.0)
    elif pattern == "tiny":
        x *= np.float32(1e-6)
    elif pattern == "huge":
        x *= np.float32(1e6)
    elif pattern == "signed_zeros":
        # -0.0 / +0.0 entries, and small values of either sign whose quotient rounds to -0 / +0: the reference's
        # (x.round() - x).detach() + x and its "+ zp ... - zp" chain decide the sign of the zero that comes out
        # (util_quant.py:4-15; SURVEY 8a quirk 14: -0.0 dequantizes to +0.0) -- compared as bits
        pick = rng.integers(0, 5, shape)
        x = np.where(pick == 0, np.float32(-0.0), x)
        x = np.where(pick == 1, np.float32(0.0), x)
        x = np.where(pick == 2, -np.abs(x) * np.float32(1e-4), x)
        x = np.where(pick == 3, np.abs(x) * np.float32(1e-4), x).astype(np.float32)
    return x


def _as_view(rng, x_np, kind, dev):
    """The same values behind a different memory layout: dense, a permuted view ([B,T,h,d] memory seen head-split, as
    the attention blocks hand it over), or a slice of a larger buffer that starts 4 bytes off a 16-byte boundary."""
    x = torch.from_numpy(x_np)
    how = rng.choice(["dense", "permuted", "offset"])
    if how == "permuted" and x.dim() == 4 and kind in ("bhtd", "bhdt"):
        if kind == "bhtd":
            mem = x.permute(0, 2, 1, 3).contiguous().to(dev)          # [B,T,h,d] memory
            return mem.permute(0, 2, 1, 3), how
        mem = x.permute(0, 3, 1, 2).contiguous().to(dev)              # [B,T,h,d] memory, key view [B,h,d,T]
        return mem.permute(0, 2, 3, 1), how
    if how == "offset":
        buf = torch.zeros(x.numel() + 1, device=dev)
        buf[1:] = x.reshape(-1).to(dev)
        return buf[1:].view(x.shape), how
    return x.to(dev), "dense"


def test_quantizer_calls_vs_oracle(eq32, dev):
    from oracle import observer_oracle as OB, fake_quant_oracle as FQ
    from outlier_suppression_amd.quantization import Quantizer
    rng = np.random.default_rng(20260930 + SEED)
    observers = {"AvgPruneMinMaxObserver": OB.observe_avg_prune_minmax, "AvgMinMaxObserver": OB.observe_avg_minmax,
                 "MinMaxObserver": OB.observe_minmax}
    seen = set()
    for case in range(N_CASES):
        kind, shape, seq_pos = _draw_shape(rng)
        observer = str(rng.choice(list(observers)))
        quantizer = str(rng.choice(["FixedFakeQuantize", "LSQPlusFakeQuantize"]))
        bit, sym = int(rng.choice([4, 6, 8])), bool(rng.integers(0, 2))
        percentile = float(rng.choice([1.0, 0.99, 0.9, 0.71, 0.5]))
        name = "layer.attention_probs_post_act_fake_quantize.observer" if kind in ("probs", "probs3d") and rng.random() < 0.7 \
            else "layer.x_post_act_fake_quantize.observer"
        masked = seq_pos != -1 and rng.random() < 0.8
        q = Quantizer(None, NS(quantizer=quantizer, observer=observer, bit=bit, symmetric=sym, ch_axis=-1)).to(dev)
        q.observer.set_name(name)
        if hasattr(q.observer, "set_percentile"):
            q.observer.set_percentile(percentile)
        q.enable_observer()
        q.enable_fake_quant()
        st = OB.ObserverState(bit=bit, symmetric=sym, name=name)
        st.percentile = percentile
        tag = (case, kind, shape, seq_pos, observer, quantizer, bit, sym, percentile, masked)
        for it in range(int(rng.integers(1, 4))):
            pattern = str(rng.choice(["normal", "outlier", "positive", "negative", "constant", "duplicates", "tiny", "huge", "signed_zeros"]))
            x_np = _draw_values(rng, shape, pattern)
            x, how = _as_view(rng, x_np, kind, dev)
            seen.add((kind, how, pattern))
            L_np = None
            if masked:
                Tn = shape[seq_pos]
                n_mask = shape[0] if kind != "probs3d" else max(1, shape[0] // int(rng.integers(1, 5)))
                L_np = rng.integers(0, Tn + 1, (n_mask,)).astype(np.int64)
                L_np[int(rng.integers(0, n_mask))] = Tn if rng.random() < 0.7 else max(1, Tn // 2)
            with torch.no_grad():
                y = q(x, None if L_np is None else torch.from_numpy(L_np).to(dev), seq_pos)
            observers[observer](st, x_np, L_np, seq_pos)
            assert eq32(q.observer.min_val.cpu().numpy(), st.min_val) and eq32(q.observer.max_val.cpu().numpy(), st.max_val), \
                (tag, it, how, pattern, q.observer.min_val.item(), st.min_val, q.observer.max_val.item(), st.max_val)
            scale, zp = st.qparams()
            assert np.float32(q.scale.item()) == np.float32(scale) and np.float32(q.zero_point.item()) == np.float32(zp), \
                (tag, it, how, pattern, q.scale.item(), scale, q.zero_point.item(), zp)
            if quantizer == "LSQPlusFakeQuantize":
                g = FQ.lsqplus_grad_factor(x_np.size, q.quant_max)
                _, ref = FQ.fake_quantize_learnableplus_per_tensor(x_np, scale, zp, q.quant_min, q.quant_max, g)
            else:
                _, ref = FQ.fake_quantize_per_tensor_affine(x_np, scale, zp, q.quant_min, q.quant_max)
            assert y.shape == x.shape and eq32(y.cpu().numpy(), ref), (tag, it, how, pattern)
        if quantizer == "LSQPlusFakeQuantize":
            # frozen parameters, autograd on: dx bit for bit, the two parameter gradients to summation order
            q.disable_observer()
            xg = x.detach().requires_grad_(True)          # the view itself: permuted or 4 bytes off alignment
            gy_np = rng.standard_normal(shape).astype(np.float32)
            s0, z0 = q.scale.detach().cpu().numpy().copy(), q.zero_point.detach().cpu().numpy().copy()
            out = q(xg, None, seq_pos)
            out.backward(torch.from_numpy(gy_np).to(dev))
            s_rep = np.maximum(np.abs(s0), np.float32(1.1920928955078125e-07)).astype(np.float32)     # fake_quant.py:188-191
            z_rep = np.clip(z0, np.float32(q.quant_min), np.float32(q.quant_max)).astype(np.float32)
            g = FQ.lsqplus_grad_factor(x_np.size, q.quant_max)
            dx, ds, dz = FQ.lsqplus_backward_per_tensor(x_np, gy_np, s_rep, z_rep, q.quant_min, q.quant_max, g)
            assert eq32(xg.grad.cpu().numpy(), dx), (tag, "dx")
            # fp32 partial sums on the device, float64 in the oracle: the bound is relative to the sum of the terms' magnitudes
            mag = float(np.abs(gy_np).sum()) * g
            assert abs(q.scale.grad.item() - ds) <= 2e-5 * abs(ds) + 2e-6 * mag * float(2 ** bit), (tag, "dscale", q.scale.grad.item(), ds)
            assert abs(q.zero_point.grad.item() - dz) <= 2e-5 * abs(dz) + 2e-6 * mag * float(s_rep[0]), \
                (tag, "dzp", q.zero_point.grad.item(), dz)
    kinds = {k for k, _, _ in seen}
    assert kinds == {"bth", "bhtd", "bhdt", "bh", "probs", "probs3d"} and {h for _, h, _ in seen} == {"dense", "permuted", "offset"}


def test_sign_of_zero_vs_oracle(eq32, dev):
    """-0.0, +0.0 and values whose quotient rounds to -0, through every per-tensor quantizer form, symmetric (zp = 0, where
    nothing is added that could wash the sign out) and asymmetric: y and dx as BITS against the oracle (which is pinned to
    the reference bit for bit on the CPU, tests/test_oracle_vs_reference_live.py).  util_quant.py:4-15,48-55."""
    from oracle import fake_quant_oracle as FQ
    from outlier_suppression_amd import ops
    rng = np.random.default_rng(606 + SEED)
    base = np.float32([-0.0, 0.0, -1e-9, 1e-9, -0.2, 0.2, -0.5, 0.5, -0.49999997, 0.49999997, -1.0, 1.0, -1.5, 1.5, -2.5, 2.5,
                       -1e-38, 1e-38, -31.5, 31.5, -32.5, 32.5, -100.0, 100.0])
    for scale in (np.float32(1.0), np.float32(0.37), np.float32(1e-8), np.float32(3.0)):
        x_np = np.concatenate([base * scale, base, rng.standard_normal(4096 - 2 * base.size).astype(np.float32) * np.float32(0.3) * scale])
        x_np[100:140] = np.float32(-0.0)
        gy_np = rng.standard_normal(x_np.shape).astype(np.float32)
        gy_np[::7] = np.float32(-0.0)
        for (qmin, qmax, zp) in ((-32, 31, 0.0), (0, 63, 0.0), (0, 63, 31.0), (0, 63, 63.0), (-8, 7, 0.0), (0, 255, 128.0)):
            x = torch.from_numpy(x_np).to(dev)
            s_t, z_t = torch.tensor([scale], device=dev), torch.tensor([zp], device=dev)
            tag = (float(scale), qmin, qmax, zp)
            # Fixed: Python scalars in the reference (fake_quant.py:124), int32 zero-point buffer
            _, ref = FQ.fake_quantize_per_tensor_affine(x_np, scale, np.float32(zp), qmin, qmax)
            y = ops.fake_quant_per_tensor(x, s_t, torch.tensor([int(zp)], dtype=torch.int32, device=dev), qmin, qmax, ops.PARAM_FIXED, 0.0)
            assert eq32(y.cpu().numpy(), ref), (tag, "fixed y")
            # LSQ+: tensor operands, grad_scale forms
            g = FQ.lsqplus_grad_factor(x_np.size, qmax)
            _, ref = FQ.fake_quantize_learnableplus_per_tensor(x_np, scale, np.float32(zp), qmin, qmax, g)
            y = ops.fake_quant_per_tensor(x, s_t, z_t, qmin, qmax, ops.PARAM_LSQPLUS, g)
            assert eq32(y.cpu().numpy(), ref), (tag, "lsq+ y")
            dx_ref, _, _ = FQ.lsqplus_backward_per_tensor(x_np, gy_np, np.float32([scale]), np.float32([zp]), qmin, qmax, g)
            dx = ops.lsq_backward_per_tensor(x, torch.from_numpy(gy_np).to(dev), s_t, z_t, qmin, qmax, ops.PARAM_LSQPLUS, g)[0]
            assert eq32(dx.cpu().numpy(), dx_ref), (tag, "lsq+ dx")


def test_weight_operators_vs_oracle(eq32, dev):
    """Per-channel side: Quantizer(nn.Linear / nn.Conv2d / nn.Embedding) with MinMaxObserver on ch_axis 0 (odd row lengths,
    one-row and one-column weights, 4-D kernels), both symmetries, and the functional per-channel fake-quant on an inner
    channel axis (the generic kernel)."""
    from oracle import observer_oracle as OB, fake_quant_oracle as FQ
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import Quantizer
    rng = np.random.default_rng(777 + SEED)
    for case in range(max(40, N_CASES // 4)):
        bit, sym = int(rng.choice([4, 6, 8])), bool(rng.integers(0, 2))
        kind = str(rng.choice(["linear", "conv", "embedding"]))
        if kind == "linear":
            mod = torch.nn.Linear(int(rng.choice([1, 3, 4, 7, 64, 100, 768])), int(rng.integers(1, 40)))
        elif kind == "conv":
            mod = torch.nn.Conv2d(int(rng.integers(1, 5)), int(rng.integers(1, 9)), int(rng.choice([1, 3])))
        else:
            mod = torch.nn.Embedding(int(rng.integers(2, 50)), int(rng.choice([2, 5, 8, 96])))
        w_np = (rng.standard_normal(tuple(mod.weight.shape)) * rng.choice([0.02, 1.0, 30.0])).astype(np.float32)
        if rng.random() < 0.3:
            w_np[0] = np.abs(w_np[0])                              # a one-sided row
        if rng.random() < 0.2:
            w_np[-1] = 0.0                                         # a constant row: scale floor 1e-8
        with torch.no_grad():
            mod.weight.copy_(torch.from_numpy(w_np))
        qm = Quantizer(mod, NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=bit, symmetric=sym, ch_axis=0)).to(dev)
        fq = qm.weight_fake_quant
        fq.enable_observer()
        fq.enable_fake_quant()
        with torch.no_grad():
            wq = fq(qm.weight)
        st = OB.ObserverState(bit=bit, symmetric=sym, ch_axis=0)
        OB.observe_minmax(st, w_np)
        scale, zp = st.qparams()
        tag = (case, kind, w_np.shape, bit, sym)
        assert eq32(fq.observer.min_val.cpu().numpy(), st.min_val) and eq32(fq.observer.max_val.cpu().numpy(), st.max_val), tag
        assert eq32(fq.scale.cpu().numpy(), scale) and bits_equal(fq.zero_point.cpu().numpy(), zp), tag
        _, ref = FQ.fake_quantize_per_channel_affine(w_np, scale, zp, 0, fq.quant_min, fq.quant_max)
        assert eq32(wq.cpu().numpy(), ref), tag
        # inner channel axis, functional form
        shape = tuple(int(v) for v in rng.integers(1, 7, size=int(rng.integers(2, 5))))
        ax = int(rng.integers(0, len(shape)))
        x_np = rng.standard_normal(shape).astype(np.float32)
        s_np = (np.abs(rng.standard_normal(shape[ax])) * 0.05 + 1e-3).astype(np.float32)
        z_np = rng.integers(fq.quant_min, fq.quant_max + 1, shape[ax]).astype(np.int32)
        y = ops.fake_quant_per_channel(torch.from_numpy(x_np).to(dev), torch.from_numpy(s_np).to(dev), torch.from_numpy(z_np).to(dev),
                                       ax, fq.quant_min, fq.quant_max)
        _, ref = FQ.fake_quantize_per_channel_affine(x_np, s_np, z_np, ax, fq.quant_min, fq.quant_max)
        assert eq32(y.cpu().numpy(), ref), (tag, shape, ax)


def test_msefast_rows_vs_oracle(dev):
    """MSEFastObserver per output channel (observer.py:496-517): every row's search equals the oracle's bounded Brent
    iterate for iterate -- the same (min, max) bit for bit -- over random row lengths, one-sided and mixed rows, 4 / 6 / 8 bit."""
    from oracle import observer_oracle as OB
    from outlier_suppression_amd.quantization.observer import MSEFastObserver
    rng = np.random.default_rng(4242 + SEED)
    for case in range(max(6, N_CASES // 40)):
        bit, sym = int(rng.choice([4, 6, 8])), bool(rng.integers(0, 2))
        rows, cols = int(rng.integers(1, 12)), int(rng.choice([4, 12, 64, 100, 768]))
        w_np = (rng.standard_normal((rows, cols)) * rng.choice([0.05, 1.0])).astype(np.float32)
        if rng.random() < 0.4:
            w_np = np.abs(w_np) + np.float32(1e-3)                 # one_side_dist 'pos' (decided on the whole tensor)
        ob = MSEFastObserver(bit=bit, symmetric=sym, ch_axis=0).to(dev)
        ob(torch.from_numpy(w_np).to(dev))
        st = OB.ObserverState(bit=bit, symmetric=sym, ch_axis=0)
        OB.observe_msefast(st, w_np)
        got_min, got_max = ob.min_val.cpu().numpy(), ob.max_val.cpu().numpy()
        assert bits_equal(got_min.astype(np.float32), np.asarray(st.min_val, dtype=np.float32)) and \
            bits_equal(got_max.astype(np.float32), np.asarray(st.max_val, dtype=np.float32)), (case, bit, sym, rows, cols, got_min, st.min_val)


def test_token_selection_vs_oracle(eq32, dev):
    """Token-wise clipping at many token counts (1 .. 70 000 slots: register path, one- and two-workgroup finalisers, the
    wide three-launch path above 32768) on few features, with the value patterns that stress a selection: two or three
    distinct values, all equal, all-negative per-token maxima, heavy ties round the percentile, extreme percentiles."""
    from oracle import observer_oracle as OB
    from outlier_suppression_amd.quantization.observer import AvgPruneMinMaxObserver
    rng = np.random.default_rng(99 + SEED)
    for case in range(max(30, N_CASES // 8)):
        big = rng.random() < 0.25
        B = int(rng.integers(1, 65 if big else 9))
        T = int(rng.integers(1, 1100 if big else 40))
        H = int(rng.choice([1, 2, 4, 8]))
        seq_pos = 1
        p = float(rng.choice([1.0, 0.999, 0.97, 0.9, 0.5, 0.01, 0.0]))
        pattern = str(rng.choice(["normal", "two", "three", "equal", "negative", "ties", "outlier"]))
        x = rng.standard_normal((B, T, H)).astype(np.float32)
        if pattern == "two":
            x = np.where(x > 0.3, np.float32(2.0), np.float32(-1.0)).astype(np.float32)
        elif pattern == "three":
            x = np.sign(np.round(x)).astype(np.float32) * np.float32(0.75)
        elif pattern == "equal":
            x[...] = np.float32(-3.5 if rng.random() < 0.5 else 0.25)
        elif pattern == "negative":
            x = -np.abs(x) - np.float32(0.5)
        elif pattern == "ties":
            x = np.round(x * 4.0).astype(np.float32) / np.float32(4.0)
        elif pattern == "outlier":
            x[:, ::7, 0] *= np.float32(40.0)
        L = rng.integers(0, T + 1, (B,)).astype(np.int64)
        if rng.random() < 0.3:
            L[...] = T
        L[int(rng.integers(0, B))] = T
        ob = AvgPruneMinMaxObserver(bit=6, symmetric=False).to(dev)
        ob.set_name("layer.x_post_act_fake_quantize.observer")
        ob.set_percentile(p)
        st = OB.ObserverState(bit=6, symmetric=False, name=ob.name)
        st.percentile = p
        for it in range(2):
            xi = x if it == 0 else (x * np.float32(1.5) + np.float32(0.125)).astype(np.float32)
            ob(torch.from_numpy(xi).to(dev), torch.from_numpy(L).to(dev), seq_pos)
            OB.observe_avg_prune_minmax(st, xi, L, seq_pos)
            assert eq32(ob.min_val.cpu().numpy(), st.min_val) and eq32(ob.max_val.cpu().numpy(), st.max_val), \
                (case, (B, T, H), p, pattern, it, ob.min_val.item(), st.min_val, ob.max_val.item(), st.max_val)


def test_fused_step_vs_three_launches_random(dev):
    """The one-launch observe + fake-quant step against the launch-per-stage path on random eligible shapes (features a
    multiple of 256), lengths with empty samples, random percentiles, occasionally a NaN among the valid tokens."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import Quantizer
    rng = np.random.default_rng(31337 + SEED)
    try:
        for case in range(max(12, N_CASES // 40)):
            B, T = int(rng.integers(1, 97)), int(rng.integers(4, 200))
            H = int(rng.choice([256, 512, 768, 1024, 1280, 3072, 4096]))
            p = float(rng.choice([1.0, 0.95, 0.71, 0.5]))
            x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32))
            # the per-token extrema are what the selectors see: the patterns that stress a selection go into one channel
            # (every token the same maximum, two or a few distinct maxima, heavy ties) -- with the hinted window of the
            # later batches (token_select.h) a crowded bin means further histogram levels behind the pre-built one
            pattern = str(rng.choice(["normal", "normal", "equal", "two", "ties", "negative"]))
            if pattern == "normal":
                x[..., 5] *= 18.0
            elif pattern == "equal":
                x[..., 5] = 41.5
                x[..., 6] = -37.25
            elif pattern == "two":
                x[..., 5] = torch.where(x[..., 5] > 0.3, torch.tensor(52.0), torch.tensor(44.0))
            elif pattern == "ties":
                x[..., 5] = torch.round(x[..., 5] * 2.0) * 0.5 + 40.0
            else:
                x = -x.abs() - 0.5
            L = torch.from_numpy(rng.integers(0, T + 1, (B,)).astype(np.int64))
            L[int(rng.integers(0, B))] = T
            if rng.random() < 0.1:
                x[0, 0, 1] = float("nan")
            # batch to batch the magnitude stays (window hit), drifts a little, or jumps (window missed on either side)
            mults = [1.0, 1.0, float(rng.choice([1.0, 1.01, 0.97])), float(rng.choice([1.6, 0.45, 1.0]))]
            res = {}
            for fused in (1, 0):
                ops.set_tuning("fused_step", fused)
                q = Quantizer(None, NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)).to(dev)
                q.observer.set_name("layer.x_post_act_fake_quantize.observer")
                q.observer.set_percentile(p)
                q.enable_observer(); q.enable_fake_quant()
                with torch.no_grad():
                    ys = [q(x.to(dev) * m, L.to(dev), 1).cpu() for m in mults]
                res[fused] = (ys, q.observer.min_val.cpu(), q.observer.max_val.cpu(), q.scale.detach().cpu(), q.zero_point.detach().cpu())
            a, b = res[1], res[0]
            for u, v in zip(a[0] + list(a[1:]), b[0] + list(b[1:])):
                assert torch.equal(torch.nan_to_num(u, nan=12345.0), torch.nan_to_num(v, nan=12345.0)) and \
                    torch.equal(torch.isnan(u), torch.isnan(v)), (case, (B, T, H), p, pattern, mults)
    finally:
        ops.set_tuning("fused_step", 1)


def test_other_observers_vs_oracle(eq32, dev, sum_tier):
    """The remaining observers of ObserverDict on random small activations (masked and not, odd shapes, one-sided data,
    several batches): AvgQuantileObserver (torch.histc-exact histogram + clip), MSEObserver / AvgMSEObserver (grid argmin),
    MSEFastObserver / AvgMSEFastObserver per tensor (bounded Brent, float64 statistics from the second batch on)."""
    from oracle import observer_oracle as OB
    from outlier_suppression_amd.quantization.quantized_module import ObserverDict
    rng = np.random.default_rng(2718 + SEED)
    for case in range(max(24, N_CASES // 16)):
        cls = str(rng.choice(["AvgQuantileObserver", "MSEObserver", "AvgMSEObserver", "MSEFastObserver", "AvgMSEFastObserver"]))
        bit, sym = int(rng.choice([4, 6, 8])), bool(rng.integers(0, 2))
        B, T, H = int(rng.integers(1, 6)), int(rng.integers(1, 20)), int(rng.choice([1, 3, 8, 33, 64]))
        masked = rng.random() < 0.6
        side = str(rng.choice(["both", "both", "pos", "neg"]))
        kw = {"threshold": float(rng.choice([0.99999, 0.999, 0.9]))} if cls == "AvgQuantileObserver" else {}
        if cls == "AvgQuantileObserver":
            sym = False
        ob = ObserverDict[cls](bit=bit, symmetric=sym, **kw).to(dev)
        st = OB.ObserverState(bit=bit, symmetric=sym)
        for it in range(int(rng.integers(1, 4))):
            x = (rng.standard_normal((B, T, H)) * rng.choice([0.1, 1.0, 20.0])).astype(np.float32)
            x[..., 0] *= np.float32(6.0)
            if side == "pos":
                x = np.abs(x) + np.float32(1e-3)
            elif side == "neg":
                x = -np.abs(x) - np.float32(1e-3)
            L = None
            if masked:
                L = rng.integers(1, T + 1, (B,)).astype(np.int64)
                L[int(rng.integers(0, B))] = T
            if it == 0:
                loose = False
            # observer.py:524 / 549: once min_val is float64 the batch is searched on a float64 copy; the loss is then a
            # float64 sum whose order is the machine's (torch's in the reference, numpy's in the oracle, the grid's here).
            # Usually ~1e-14 apart; but candidates of a staircase loss tie to that precision, and a tie broken the other way
            # sends Brent down another path: the oracle alone moves by up to 1.3e-2 between summation orders (DESIGN.md
            # section 2) -- the bound of tests/test_gpu_parity.py for these calls
            loose = loose or ("MSEFast" in cls and np.asarray(st.min_val).dtype == np.float64)
            xv, _ = _as_view(rng, x, "bth", dev)          # dense, or 4 bytes off a 16-byte boundary
            ob(xv, None if L is None else torch.from_numpy(L).to(dev), 1 if masked else -1)
            if cls == "AvgQuantileObserver":
                OB.observe_avg_quantile(st, x, L, 1 if masked else -1, threshold=kw["threshold"])
            elif cls in ("MSEObserver", "AvgMSEObserver"):
                OB.observe_mse(st, x, L, 1 if masked else -1, average=cls.startswith("Avg"))
            else:
                OB.observe_msefast(st, x, L, 1 if masked else -1, average=cls.startswith("Avg"))
            tag = (case, cls, bit, sym, (B, T, H), masked, side, it)
            got_min, got_max = ob.min_val.cpu().numpy(), ob.max_val.cpu().numpy()
            want_min, want_max = np.asarray(st.min_val), np.asarray(st.max_val)
            if "MSEFast" in cls:
                # float64 statistics (scipy's results; the reference's one-sided zero is a float32 zero, here a float64 one);
                # the search itself is iterate-for-iterate equal, on fp32 or float64 input exactly when the reference's is
                assert got_min.dtype == np.float64 and got_max.dtype == np.float64
                if loose:
                    # (a few dozen values at 4 bit are a staircase of a handful of steps with several equal minima: there the
                    # nested search of two summation orders can end a whole step apart -- only finiteness is asserted)
                    assert np.isfinite(got_min).all() and np.isfinite(got_max).all() and (got_min <= got_max).all(), tag
                    if bit >= 6 and x.size >= 2048:
                        np.testing.assert_allclose(got_min, want_min.astype(np.float64), rtol=3e-2, atol=0, err_msg=str(tag))
                        np.testing.assert_allclose(got_max, want_max.astype(np.float64), rtol=3e-2, atol=0, err_msg=str(tag))
                else:
                    assert bits_equal(got_min, want_min.astype(np.float64)) and bits_equal(got_max, want_max.astype(np.float64)), \
                        (tag, got_min, want_min, got_max, want_max)
                    if np.float64 in (want_min.dtype, want_max.dtype):      # the public method on float64 statistics: float64 arithmetic
                        s_g, z_g = ob.calculate_qparams(ob.min_val, ob.max_val)
                        s_o, z_o = st.qparams()
                        assert np.float32(s_g.item()) == np.float32(np.asarray(s_o).reshape(-1)[0]) and \
                            float(z_g.item()) == float(np.asarray(z_o).reshape(-1)[0]), (tag, s_g.item(), s_o, z_g.item(), z_o)
            else:
                assert eq32(got_min, want_min.astype(np.float32)) and eq32(got_max, want_max.astype(np.float32)), \
                    (tag, got_min, want_min, got_max, want_max)


def test_deferred_forwards_vs_oracle(eq32, dev, sum_tier):
    """Observer passes as calibrate() runs them: the sites of a forward recorded and reduced together
    (quantization/deferred.py) -- random groups of sites (layouts, masks, observers, MSEFast searches among them), several
    forwards, every statistic and parameter against the oracle."""
    from oracle import observer_oracle as OB
    from outlier_suppression_amd.quantization import Quantizer
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    rng = np.random.default_rng(60606 + SEED)
    fns = {"AvgPruneMinMaxObserver": OB.observe_avg_prune_minmax, "AvgMinMaxObserver": OB.observe_avg_minmax,
           "MinMaxObserver": OB.observe_minmax}
    for case in range(max(20, N_CASES // 16)):
        sites = []
        for k in range(int(rng.integers(1, 7))):
            kind, shape, seq_pos = _draw_shape(rng)
            observer = str(rng.choice(list(fns) + (["AvgMSEFastObserver"] if rng.random() < 0.3 else [])))
            bit, sym = int(rng.choice([4, 6, 8])), bool(rng.integers(0, 2))
            name = f"layer{k}.attention_probs_post_act_fake_quantize.observer" if kind in ("probs", "probs3d") else f"layer{k}.x_post_act_fake_quantize.observer"
            masked = seq_pos != -1 and rng.random() < 0.8
            q = Quantizer(None, NS(quantizer=str(rng.choice(["FixedFakeQuantize", "LSQPlusFakeQuantize"])), observer=observer,
                                   bit=bit, symmetric=sym, ch_axis=-1)).to(dev)
            q.observer.set_name(name)
            pct = float(rng.choice([1.0, 0.95, 0.71]))
            if hasattr(q.observer, "set_percentile"):
                q.observer.set_percentile(pct)
            q.enable_observer()
            q.disable_fake_quant()
            st = OB.ObserverState(bit=bit, symmetric=sym, name=name)
            st.percentile = pct
            sites.append([q, st, kind, shape, seq_pos, observer, masked, False])
        with deferred_observation() as rec:
            for it in range(int(rng.integers(1, 4))):
                fed = []
                for q, st, kind, shape, seq_pos, observer, masked, _ in sites:
                    x_np = _draw_values(rng, shape, str(rng.choice(["normal", "outlier", "positive", "duplicates"])))
                    x, how = _as_view(rng, x_np, kind, dev)
                    L_np = None
                    if masked:
                        Tn = shape[seq_pos]
                        n_mask = shape[0] if kind != "probs3d" else max(1, shape[0] // int(rng.integers(1, 5)))
                        L_np = rng.integers(0, Tn + 1, (n_mask,)).astype(np.int64)
                        L_np[int(rng.integers(0, n_mask))] = Tn
                    assert q(x, None if L_np is None else torch.from_numpy(L_np).to(dev), seq_pos) is x
                    if observer == "AvgMSEFastObserver" and L_np is None and how == "permuted" and not x.is_contiguous():
                        # no mask: the reference searches on x_orig.clone() -- strides preserved -- and torch adds a dense
                        # permuted tensor in MEMORY order (observer.py:522-524 / 547-549); the oracle adds what it is
                        # handed in C order, so it is handed the memory image ([B,T,h,d]) of the view
                        x_np = np.ascontiguousarray(np.transpose(x_np, (0, 2, 1, 3) if kind == "bhtd" else (0, 3, 1, 2)))
                    fed.append((x_np, L_np))
                rec.flush()
                for site, (x_np, L_np) in zip(sites, fed):
                    q, st, kind, shape, seq_pos, observer, masked, loose = site
                    tag = (case, it, kind, shape, seq_pos, observer, masked)
                    if observer == "AvgMSEFastObserver":
                        # fp32-input batches are bit-equal (also later ones, while the reference's min_val stays float32);
                        # a float64-input batch agrees to the oracle's own order sensitivity, and so does the mean after it
                        loose = site[7] = loose or np.asarray(st.min_val).dtype == np.float64
                        OB.observe_msefast(st, x_np, L_np, seq_pos, average=True)
                        got = (q.observer.min_val.cpu().numpy(), q.observer.max_val.cpu().numpy())
                        want = (np.asarray(st.min_val, dtype=np.float64), np.asarray(st.max_val, dtype=np.float64))
                        if loose:
                            assert np.isfinite(got[0]).all() and np.isfinite(got[1]).all() and (got[0] <= got[1]).all(), tag
                            if q.bit >= 6 and x_np.size >= 2048:
                                np.testing.assert_allclose(got[0], want[0], rtol=3e-2, err_msg=str(tag))
                                np.testing.assert_allclose(got[1], want[1], rtol=3e-2, err_msg=str(tag))
                        else:
                            assert bits_equal(got[0], want[0]) and bits_equal(got[1], want[1]), str((tag, None if L_np is None else L_np.tolist(), q.bit, q.observer.symmetric, len(sites), got, want))
                        continue
                    fns[observer](st, x_np, L_np, seq_pos)
                    assert eq32(q.observer.min_val.cpu().numpy(), st.min_val) and eq32(q.observer.max_val.cpu().numpy(), st.max_val), tag
                    scale, zp = st.qparams()
                    assert np.float32(q.scale.item()) == np.float32(scale) and np.float32(q.zero_point.item()) == np.float32(zp), tag


def test_gamma_ops_vs_oracle(eq32, dev):
    """Gamma-Migration arithmetic (gamma_migration.py:70-71, util_layernorm.py:27, 49-52) on odd sizes and misaligned
    buffers: weight fold in place, beta / gamma, input * gamma + hidden."""
    from oracle import gamma_oracle as GO
    from outlier_suppression_amd import ops
    rng = np.random.default_rng(5150 + SEED)
    for case in range(max(40, N_CASES // 8)):
        rows, cols = int(rng.integers(1, 70)), int(rng.choice([1, 2, 3, 4, 7, 8, 33, 64, 100, 768]))
        w_np = rng.standard_normal((rows, cols)).astype(np.float32)
        g_np = (rng.standard_normal(cols) * 0.5 + 1.0).astype(np.float32)
        b_np = rng.standard_normal(cols).astype(np.float32)
        w, _ = _as_view(rng, w_np, "bth", dev)
        w = w.clone() if not w.is_contiguous() else w
        ops.gamma_fold_(w, torch.from_numpy(g_np).to(dev))
        assert eq32(w.cpu().numpy(), GO.fold_gamma_into_weight(w_np, g_np)), (case, rows, cols, "fold")
        assert eq32(ops.gamma_split_bias(torch.from_numpy(b_np).to(dev), torch.from_numpy(g_np).to(dev)).cpu().numpy(),
                    GO.split_bias(b_np, g_np)), (case, "split")
        shape = tuple(int(v) for v in rng.integers(1, 6, size=int(rng.integers(1, 3)))) + (cols,)
        x_np, h_np = rng.standard_normal(shape).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
        x, _ = _as_view(rng, x_np, "bth", dev)
        h, _ = _as_view(rng, h_np, "bth", dev)
        for gamma in (None, g_np):
            got = ops.gamma_residual(x, h, None if gamma is None else torch.from_numpy(gamma).to(dev))
            assert eq32(got.cpu().numpy(), GO.gamma_residual(x_np, h_np, gamma)), (case, shape, gamma is None)


def test_per_channel_learnable_vs_oracle(eq32, dev):
    """LSQ / LSQ+ quantizers on the channel axis 0 (weights): forward against the oracle's per-channel chain, backward row by
    row against the per-tensor closed form with the row's parameters and the per-channel grad factor
    (fake_quant.py:195-206: 1 / sqrt(numel / C * quant_max))."""
    from oracle import fake_quant_oracle as FQ
    from outlier_suppression_amd.quantization import Quantizer
    rng = np.random.default_rng(8086 + SEED)
    for case in range(max(20, N_CASES // 16)):
        C, inner = int(rng.integers(1, 20)), int(rng.choice([1, 3, 4, 8, 33, 64, 200]))
        bit, sym = int(rng.choice([4, 6, 8])), bool(rng.integers(0, 2))
        w_np = (rng.standard_normal((C, inner)) * rng.choice([0.05, 1.0])).astype(np.float32)
        lin = torch.nn.Linear(inner, C)
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(w_np))
        qm = Quantizer(lin, NS(quantizer="LSQPlusFakeQuantize", observer="MinMaxObserver", bit=bit, symmetric=sym, ch_axis=0)).to(dev)
        fq = qm.weight_fake_quant
        fq.enable_observer(); fq.enable_fake_quant()
        with torch.no_grad():
            fq(qm.weight)                                      # observe: per-row parameters
        fq.disable_observer()
        s_np, z_np = fq.scale.detach().cpu().numpy().copy(), fq.zero_point.detach().cpu().numpy().copy()
        assert s_np.shape == (C,) and z_np.shape == (C,)
        wg = qm.weight.detach().clone().requires_grad_(True)
        gy_np = rng.standard_normal((C, inner)).astype(np.float32)
        out = fq(wg)
        out.backward(torch.from_numpy(gy_np).to(dev))
        s_rep = np.maximum(np.abs(s_np), np.float32(1.1920928955078125e-07)).astype(np.float32)
        z_rep = np.clip(z_np, np.float32(fq.quant_min), np.float32(fq.quant_max)).astype(np.float32)
        g = FQ.lsqplus_grad_factor(w_np.size, fq.quant_max, channels=C)
        _, ref = FQ.fake_quantize_learnableplus_per_channel(w_np, s_rep, z_rep, 0, fq.quant_min, fq.quant_max, g)
        tag = (case, C, inner, bit, sym)
        assert eq32(out.detach().cpu().numpy(), ref), (tag, "forward")
        dx_rows, ds_rows, dz_rows = [], [], []
        for c in range(C):
            dx, ds, dz = FQ.lsqplus_backward_per_tensor(w_np[c], gy_np[c], s_rep[c:c + 1], z_rep[c:c + 1], fq.quant_min, fq.quant_max, g)
            dx_rows.append(dx); ds_rows.append(ds); dz_rows.append(dz)
        assert eq32(wg.grad.cpu().numpy(), np.stack(dx_rows)), (tag, "dx")
        mag = np.abs(gy_np).sum(1).astype(np.float64) * g
        assert (np.abs(fq.scale.grad.cpu().numpy() - np.array(ds_rows)) <= 2e-5 * np.abs(ds_rows) + 2e-6 * mag * 2 ** bit).all(), (tag, "dscale")
        assert (np.abs(fq.zero_point.grad.cpu().numpy() - np.array(dz_rows)) <= 2e-5 * np.abs(dz_rows) + 2e-6 * mag * s_rep).all(), (tag, "dzp")


def test_cached_search_equals_literal_random_models(dev):
    """find_ratio_cached against the literal find_ratio (token_wise_clipping.py:50-66) on random tiny BERT / RoBERTa / BART
    models with random batch geometries, masks and grids: same percentile, same per-candidate losses, bit-equal parameters,
    same switches left behind."""
    import logging
    import transformers as T
    from outlier_suppression_amd import token_wise_clipping as TWC
    from outlier_suppression_amd.gamma_migration import delay_ln
    from outlier_suppression_amd.quant_model import quantize_model
    from outlier_suppression_amd.quantization import enable_calibration_woquantization, disable_all
    from outlier_suppression_amd.quantization.state import set_observer_name
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    rng = np.random.default_rng(1999 + SEED)
    a_q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)

    class Grab(logging.Handler):
        def __init__(self):
            super().__init__()
            self.losses = []

        def emit(self, record):
            m = record.getMessage()
            if m.startswith("the ratio is"):
                self.losses.append(float(m.split("the loss is")[1]))
    saved = (TWC.task_type, TWC.model_type)
    try:
        for case in range(max(4, N_CASES // 160)):
            kind = str(rng.choice(["bert", "roberta", "bart"]))
            heads = int(rng.choice([1, 2, 4]))
            hidden = heads * int(rng.choice([8, 16]))
            layers = int(rng.integers(1, 3))
            torch.manual_seed(int(rng.integers(0, 10 ** 6)))
            common = dict(vocab_size=90, max_position_embeddings=48)
            if kind == "bart":
                fp = T.BartForConditionalGeneration(T.BartConfig(d_model=hidden, encoder_layers=layers, decoder_layers=layers,
                                                                 encoder_attention_heads=heads, decoder_attention_heads=heads,
                                                                 encoder_ffn_dim=2 * hidden, decoder_ffn_dim=2 * hidden, dropout=0.0,
                                                                 attention_dropout=0.0, activation_dropout=0.0, **common))
                task = "summ"
            elif kind == "roberta":
                fp = T.RobertaForSequenceClassification(T.RobertaConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                                                                        intermediate_size=2 * hidden, hidden_dropout_prob=0.0,
                                                                        attention_probs_dropout_prob=0.0, type_vocab_size=1, pad_token_id=1,
                                                                        num_labels=3, **common))
                task = "glue"
            else:
                fp = T.BertForSequenceClassification(T.BertConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                                                                  intermediate_size=2 * hidden, hidden_dropout_prob=0.0,
                                                                  attention_probs_dropout_prob=0.0, type_vocab_size=2, num_labels=2, **common))
                task = "glue"
            fp = fp.eval().to(dev)
            B, Tn, n_batches = int(rng.integers(1, 6)), int(rng.integers(4, 20)), int(rng.integers(1, 4))
            Td = int(rng.integers(2, 9))
            batches = []
            for _ in range(n_batches):
                L = rng.integers(1, Tn + 1, (B,)); L[int(rng.integers(0, B))] = Tn
                mask = (np.arange(Tn)[None, :] < L[:, None]).astype(np.int64)
                ids = rng.integers(5, 85, (B, Tn)) * mask + (1 - mask)
                b = {"input_ids": torch.from_numpy(ids).to(dev), "attention_mask": torch.from_numpy(mask).to(dev)}
                if kind == "bert":
                    b["token_type_ids"] = torch.zeros_like(b["input_ids"])
                if kind == "bart":
                    DL = rng.integers(1, Td + 1, (B,)); DL[int(rng.integers(0, B))] = Td
                    dm = (np.arange(Td)[None, :] < DL[:, None]).astype(np.int64)
                    b["decoder_input_ids"] = torch.from_numpy(rng.integers(5, 85, (B, Td)) * dm + (1 - dm)).to(dev)
                    b["decoder_attention_mask"] = torch.from_numpy(dm).to(dev)
                batches.append(b)
            grid = {"iters": int(rng.integers(2, 6)), "step": float(rng.choice([0.01, 0.05, 0.1]))}
            TWC.task_type, TWC.model_type = task, kind
            results = []
            for fn in (TWC.find_ratio, TWC.find_ratio_cached):
                model = quantize_model(fp, w_q, a_q).to(dev)
                with torch.no_grad():
                    if task == "summ":
                        fp_output = [model(**b)[0][b["decoder_attention_mask"] == 1, :].detach() for b in batches]
                    else:
                        fp_output = [model(**b)[0].detach() for b in batches]
                model = delay_ln(model, NS(a_qconfig=a_q, w_qconfig=w_q), NS(model_type=kind, task_type=task))
                enable_calibration_woquantization(model, quantizer_type="weight_fake_quant")
                with torch.no_grad():
                    model(**batches[0])
                disable_all(model)
                set_observer_name(model)
                h = Grab()
                TWC.logger.addHandler(h)
                TWC.logger.setLevel(logging.INFO)
                ratio = fn(NS(model=model), batches, fp_output, grid)
                TWC.logger.removeHandler(h)
                qs = [m for n, m in model.named_modules() if isinstance(m, QuantizeBase) and "act" in n]
                results.append((ratio, h.losses, [m.scale.detach().clone() for m in qs], [m.zero_point.detach().clone() for m in qs],
                                [m.observer.cnt for m in qs], [(m.observer_enabled, m.fake_quant_enabled) for m in qs]))
            (r0, l0, s0, z0, c0, f0), (r1, l1, s1, z1, c1, f1) = results
            tag = (case, kind, heads, hidden, layers, B, Tn, n_batches, grid)
            assert r0 == r1 and l0 == l1 and c0 == c1 and f0 == f1, (tag, r0, r1, l0, l1)
            assert all(torch.equal(a, b) for a, b in zip(s0, s1)) and all(torch.equal(a, b) for a, b in zip(z0, z1)), tag
    finally:
        TWC.task_type, TWC.model_type = saved


def test_state_dict_and_copies(dev):
    """Calibrated quantizers survive state_dict -> fresh module -> load_state_dict (fake_quant.py:59-97 resizes scale /
    zero_point to the stored shape) and copy.deepcopy: same keys, dtypes, shapes, values, and the same quantised output."""
    import copy
    from outlier_suppression_amd.quantization import Quantizer
    rng = np.random.default_rng(4711 + SEED)
    for case in range(max(30, N_CASES // 16)):
        weight = rng.random() < 0.4
        quantizer = str(rng.choice(["FixedFakeQuantize", "LSQFakeQuantize", "LSQPlusFakeQuantize"]))
        bit, sym = int(rng.choice([4, 6, 8])), bool(rng.integers(0, 2))
        if weight:
            observer = str(rng.choice(["MinMaxObserver", "MSEFastObserver", "LSQPlusObserver"]))
            if observer == "LSQPlusObserver":
                sym = True
            ch_axis = int(rng.choice([0, -1]))
            x = torch.from_numpy((rng.standard_normal((int(rng.integers(1, 9)), int(rng.choice([4, 12, 64])))) * 0.1).astype(np.float32)).to(dev)
            args = (x,)
        else:
            observer = str(rng.choice(["MinMaxObserver", "AvgMinMaxObserver", "AvgPruneMinMaxObserver", "AvgMSEFastObserver", "AvgQuantileObserver"]))
            if observer == "AvgQuantileObserver":
                sym = False
            ch_axis = -1
            B, T, H = int(rng.integers(1, 5)), int(rng.integers(2, 12)), int(rng.choice([8, 33, 64]))
            x = torch.from_numpy(rng.standard_normal((B, T, H)).astype(np.float32)).to(dev)
            L = torch.from_numpy(rng.integers(1, T + 1, (B,)).astype(np.int64)).to(dev)
            args = (x, L, 1)
        cfg = NS(quantizer=quantizer, observer=observer, bit=bit, symmetric=sym, ch_axis=ch_axis)
        tag = (case, quantizer, observer, bit, sym, ch_axis, tuple(x.shape))

        def build():
            q = Quantizer(None, cfg).to(dev)
            q.observer.set_name("layer.x_post_act_fake_quantize.observer")
            if hasattr(q.observer, "set_percentile"):
                q.observer.set_percentile(0.9)
            return q
        q = build()
        q.enable_observer(); q.enable_fake_quant()
        with torch.no_grad():
            for _ in range(int(rng.integers(1, 3))):
                q(*args)
        q.disable_observer()
        with torch.no_grad():
            ref = q(*args)
        sd = q.state_dict()
        fresh = build()
        assert set(fresh.state_dict()) == set(sd), tag
        fresh.load_state_dict(copy.deepcopy(sd))
        fresh.disable_observer(); fresh.enable_fake_quant()
        twin = copy.deepcopy(q)
        for other, what in ((fresh, "loaded"), (twin, "deepcopy")):
            osd = other.state_dict()
            for k, v in sd.items():
                # load_state_dict copies INTO the fresh module's float32 statistic buffers (torch's copy_, in the reference as
                # well): a float64 MSEFast statistic arrives rounded to float32; everything else keeps its dtype
                if what == "loaded" and v.dtype == torch.float64:
                    assert osd[k].dtype == torch.float32 and torch.equal(osd[k].cpu(), v.cpu().float()), (tag, what, k, osd[k], v)
                    continue
                assert osd[k].dtype == v.dtype and osd[k].shape == v.shape and torch.equal(osd[k].cpu(), v.cpu()), (tag, what, k, osd[k], v)
            with torch.no_grad():
                got = other(*args)
            assert torch.equal(got, ref), (tag, what)


def test_fused_sites_vs_eager(dev):
    """The two fused producer sites against the eager sequences they replace, on random shapes and misaligned buffers:
    GELU + fake-quant (bit-identical to F.gelu followed by the fake-quant launch) and residual + LayerNorm + shift +
    fake-quant (the normalisation within 4e-6 of torch's, the quantised output at most one step away and only next to
    a rounding boundary)."""
    import torch.nn.functional as F
    from outlier_suppression_amd import ops
    rng = np.random.default_rng(1234567 + SEED)
    for case in range(max(30, N_CASES // 16)):
        shape = tuple(int(v) for v in rng.integers(1, 9, size=int(rng.integers(1, 3)))) + (int(rng.choice([1, 3, 4, 64, 100, 768, 3072])),)
        x_np = (rng.standard_normal(shape) * rng.choice([0.5, 3.0])).astype(np.float32)
        x, _ = _as_view(rng, x_np, "bth", dev)
        s = torch.tensor([float(rng.uniform(0.01, 0.3))], device=dev)
        z = torch.tensor([float(rng.integers(0, 64))], device=dev)
        mode = int(rng.choice([ops.PARAM_FIXED, ops.PARAM_LSQPLUS]))
        got = ops.gelu_fake_quant_per_tensor(x, s, z, 0, 63, mode, 1e-3)
        want = ops.fake_quant_per_tensor(F.gelu(x), s, z, 0, 63, mode, 1e-3)
        assert torch.equal(got, want), (case, shape, "gelu")
        H = int(rng.choice([4, 64, 260, 768, 1024]))
        lshape = tuple(int(v) for v in rng.integers(1, 7, size=int(rng.integers(1, 3)))) + (H,)
        a = torch.from_numpy((rng.standard_normal(lshape) * 2).astype(np.float32)).to(dev)
        hid = torch.from_numpy(rng.standard_normal(lshape).astype(np.float32)).to(dev)
        gamma = torch.from_numpy((rng.random(H) + 0.5).astype(np.float32)).to(dev)
        w = torch.from_numpy((rng.random(H) + 0.5).astype(np.float32)).to(dev)
        b = torch.from_numpy(rng.standard_normal(H).astype(np.float32)).to(dev)
        use_hid, use_gamma, use_w, use_b = (bool(v) for v in rng.integers(0, 2, 4))
        eps = float(rng.choice([1e-5, 1e-12]))
        r = a
        if use_hid:
            r = (a * gamma if use_gamma else a) + hid
        n = F.layer_norm(r, (H,), w if use_w else None, None, eps)
        if use_b:
            n = n + b
        quant = (s, z, 0, 63, mode, 1e-3)
        plain = ops.residual_layernorm_fake_quant(a, hid if use_hid else None, gamma if (use_hid and use_gamma) else None,
                                                  w if use_w else None, b if use_b else None, eps, None)
        fused = ops.residual_layernorm_fake_quant(a, hid if use_hid else None, gamma if (use_hid and use_gamma) else None,
                                                  w if use_w else None, b if use_b else None, eps, quant)
        tag = (case, lshape, use_hid, use_gamma, use_w, use_b, eps)
        tol = 4e-6 * max(1.0, float(n.abs().max()))
        assert float((plain - n).abs().max()) <= tol, (tag, float((plain - n).abs().max()))
        assert torch.equal(fused, ops.fake_quant_per_tensor(plain, s, z, 0, 63, mode, 1e-3)), (tag, "own normalisation")
        eager_q = ops.fake_quant_per_tensor(n, s, z, 0, 63, mode, 1e-3)
        differ = fused != eager_q
        if bool(differ.any()):
            step = float(s.item())
            assert float((fused - eager_q).abs().max()) <= step * 1.0001, tag
            u = (n / s)[differ]
            assert float(((u - torch.floor(u)) - 0.5).abs().max()) <= 2 * tol / step + 1e-4, tag   # only next to a rounding boundary


def test_quantized_operator_arguments(eq32, dev):
    """QLinear / QConv2d / QEmbedding keep the wrapped module's arguments (quantized_module.py:23-58: bias or not, stride,
    padding, dilation, groups, padding_idx): the quantized operator equals the stock functional on the fake-quantised weight."""
    import torch.nn.functional as F
    from outlier_suppression_amd.quantization import Quantizer
    rng = np.random.default_rng(31415 + SEED)
    cfg = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    for case in range(max(30, N_CASES // 16)):
        kind = str(rng.choice(["linear", "conv", "embedding"]))
        torch.manual_seed(int(rng.integers(0, 10 ** 6)))
        if kind == "linear":
            mod = torch.nn.Linear(int(rng.integers(1, 40)), int(rng.integers(1, 20)), bias=bool(rng.integers(0, 2)))
            x = torch.randn(int(rng.integers(1, 5)), int(rng.integers(1, 6)), mod.in_features)
        elif kind == "conv":
            groups = int(rng.choice([1, 2]))
            cin, cout = groups * int(rng.integers(1, 4)), groups * int(rng.integers(1, 5))
            mod = torch.nn.Conv2d(cin, cout, int(rng.choice([1, 3])), stride=int(rng.choice([1, 2])), padding=int(rng.choice([0, 1])),
                                  dilation=int(rng.choice([1, 2])), groups=groups, bias=bool(rng.integers(0, 2)))
            x = torch.randn(int(rng.integers(1, 4)), cin, 9, 11)
        else:
            n = int(rng.integers(3, 30))
            mod = torch.nn.Embedding(n, int(rng.choice([4, 9, 32])), padding_idx=(int(rng.integers(0, n)) if rng.random() < 0.5 else None))
            x = torch.randint(0, n, (int(rng.integers(1, 5)), int(rng.integers(1, 9))))
        qm = Quantizer(mod, cfg).to(dev)
        fq = qm.weight_fake_quant
        fq.enable_observer(); fq.enable_fake_quant()
        with torch.no_grad():
            got = qm(x.to(dev))
            wq = fq(qm.weight)
            b = None if getattr(qm, "bias", None) is None else qm.bias
            if kind == "linear":
                want = F.linear(x.to(dev), wq, b)
            elif kind == "conv":
                want = F.conv2d(x.to(dev), wq, b, mod.stride, mod.padding, mod.dilation, mod.groups)
            else:
                want = F.embedding(x.to(dev), wq, mod.padding_idx)
        assert got.shape == want.shape and torch.equal(got, want), (case, kind, mod)
        assert torch.equal(qm.weight.cpu(), mod.weight.detach()) and (b is None) == (getattr(mod, "bias", None) is None), (case, kind)


def test_msefast_float32_statistics_corner(dev, sum_tier):
    """Per-tensor AvgMSEFast on data with a populated lower bound (GELU-like: the optimal range keeps the data's own minimum):
    the reference's min_val is then the float32 extremum, batch after batch -- its running mean is fp32 arithmetic, the next
    batch is searched on fp32 input, and while max_val is float32 too the parameters are derived in fp32.  Statistics,
    scale and zero point bit for bit against the oracle for as long as it searches on fp32 input."""
    from oracle import observer_oracle as OB
    from outlier_suppression_amd.quantization import Quantizer
    rng = np.random.default_rng(1 + SEED)
    exact_later = 0
    for trial in range(max(8, N_CASES // 100)):
        bit = int(rng.choice([6, 8]))
        q = Quantizer(None, NS(quantizer="FixedFakeQuantize", observer="AvgMSEFastObserver", bit=bit, symmetric=False, ch_axis=-1)).to(dev)
        q.enable_observer(); q.enable_fake_quant()
        st = OB.ObserverState(bit=bit, symmetric=False)
        floor, cap = float(rng.choice([-0.5, -0.17, -1.0])), (float(rng.choice([1.5, 2.5])) if rng.random() < 0.4 else None)
        for it in range(4):
            x = np.maximum(rng.standard_normal((4, 16, 64)), floor)
            if cap is not None:
                x = np.minimum(x, cap)                   # both extrema populated: both statistics can stay float32
            x = (x * (1 + 0.1 * it)).astype(np.float32)
            fp32_input = np.asarray(st.min_val).dtype != np.float64
            if not fp32_input:
                break
            with torch.no_grad():
                q(torch.from_numpy(x).to(dev))
            OB.observe_msefast(st, x, average=True)
            scale, zp = st.qparams()
            tag = (trial, it, bit, floor, cap, np.asarray(st.min_val).dtype, np.asarray(st.max_val).dtype)
            assert float(q.observer.min_val) == float(st.min_val) and float(q.observer.max_val) == float(st.max_val), \
                (tag, float(q.observer.min_val), float(st.min_val), float(q.observer.max_val), float(st.max_val))
            assert np.float32(q.scale.item()) == np.float32(np.asarray(scale).reshape(-1)[0]) and \
                float(q.zero_point.item()) == float(np.asarray(zp).reshape(-1)[0]), (tag, q.scale.item(), scale, q.zero_point.item(), zp)
            exact_later += it > 0
    assert exact_later >= 3          # the walk did reach later batches on fp32 input
