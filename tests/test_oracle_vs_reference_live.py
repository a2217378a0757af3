"""CPU, build container only: the NumPy oracle against the REFERENCE ITSELF on freshly drawn inputs.

The committed goldens (tests/golden/*.npz) pin the oracle on fixed vectors; here the reference's own functions
and observer classes are imported (two process-local shims: empty ``seaborn`` module, identity ``Tensor.cuda``)
and run next to the oracle on seeded random cases that no fixture holds -- shapes, bit widths, masks, percentiles,
one-sided ranges.  Bit-exact on everything.  Skipped where /root/reference is absent (the GPU box).
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not available")
F32 = np.float32


@pytest.fixture(scope="module")
def ref():
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from quant_transformer.quantization import observer, util_quant
    torch.set_num_threads(1)
    return observer, util_quant


def _bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=F32)).view(np.uint32)


def test_fake_quant_random_cases(ref):
    _, U = ref
    from oracle import fake_quant_oracle as FQ
    gen = torch.Generator().manual_seed(20260930)
    for trial in range(60):
        bit = int(torch.randint(2, 9, (1,), generator=gen))
        sym = bool(torch.randint(0, 2, (1,), generator=gen))
        qmin, qmax = (-(1 << (bit - 1)), (1 << (bit - 1)) - 1) if sym else (0, (1 << bit) - 1)
        shape = tuple(int(v) for v in torch.randint(1, 9, (int(torch.randint(1, 4, (1,), generator=gen)),), generator=gen))
        x = torch.randn(*shape, generator=gen) * float(10 ** torch.empty(1).uniform_(-3, 2, generator=gen))
        scale = float(torch.empty(1).uniform_(1e-4, 2.0, generator=gen))
        zp = 0 if sym else int(torch.randint(qmin, qmax + 1, (1,), generator=gen))
        want = U.fake_quantize_per_tensor_affine(x, scale, zp, qmin, qmax)
        _, got = FQ.fake_quantize_per_tensor_affine(x.numpy(), F32(scale), F32(zp), qmin, qmax)
        assert np.array_equal(_bits(got), _bits(want.numpy())), (trial, bit, sym, shape)
        if x.dim() >= 2:
            C = x.shape[0]
            s = torch.empty(C).uniform_(1e-3, 1.0, generator=gen)
            z = torch.zeros(C, dtype=torch.int32) if sym else torch.randint(qmin, qmax + 1, (C,), generator=gen).int()
            want = U.fake_quantize_per_channel_affine(x, s, z, 0, qmin, qmax)
            _, got = FQ.fake_quantize_per_channel_affine(x.numpy(), s.numpy(), z.numpy(), 0, qmin, qmax)
            assert np.array_equal(_bits(got), _bits(want.numpy())), (trial, "per-channel")


def test_observers_random_sequences(ref):
    O, _ = ref
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(77001)
    table = [("MinMaxObserver", OB.observe_minmax), ("AvgMinMaxObserver", OB.observe_avg_minmax),
             ("AvgPruneMinMaxObserver", OB.observe_avg_prune_minmax)]
    for trial in range(45):
        cls_name, fn = table[trial % 3]
        bit = int(torch.randint(3, 9, (1,), generator=gen))
        sym = bool(torch.randint(0, 2, (1,), generator=gen))
        B, T, H = (int(torch.randint(lo, hi, (1,), generator=gen)) for lo, hi in ((1, 7), (2, 12), (2, 10)))
        layout = trial % 4
        ob = getattr(O, cls_name)(bit=bit, symmetric=sym, ch_axis=-1)
        name = "m.attention_probs_x" if trial % 9 == 8 else "m.x_post_act_fake_quantize"
        ob.set_name(name)
        st = OB.ObserverState(bit=bit, symmetric=sym, name=name)
        pct = float(torch.empty(1).uniform_(0.05, 1.0, generator=gen))
        if cls_name == "AvgPruneMinMaxObserver":
            ob.set_percentile(pct)
            st.percentile = pct
        for it in range(3):
            shift = float(torch.empty(1).uniform_(-3, 3, generator=gen)) if trial % 5 == 0 else 0.0
            if layout == 0:
                x, sp = torch.randn(B, T, H, generator=gen) + shift, 1
            elif layout == 1:
                x, sp = torch.randn(B, 2, T, H, generator=gen) + shift, 2
            elif layout == 2:
                x, sp = (torch.randn(B, T, 2, H, generator=gen) + shift).permute(0, 2, 3, 1), 3
            else:
                x, sp = torch.randn(B, T, H, generator=gen).abs() + 0.1, 1       # one-sided range
            L = torch.randint(1, T + 1, (B,), generator=gen)
            masked = trial % 7 != 6
            if masked:
                ob(x, L, sp)
                fn(st, x.numpy(), L.numpy(), sp)
            elif cls_name == "AvgPruneMinMaxObserver":
                ob(x, None, sp)                      # no mask, seq_pos given: reshape_batch_embedding path
                fn(st, x.numpy(), None, sp)
            else:
                ob(x)
                fn(st, x.numpy())
            assert np.array_equal(_bits(st.min_val), _bits(ob.min_val.numpy())), (trial, cls_name, it)
            assert np.array_equal(_bits(st.max_val), _bits(ob.max_val.numpy())), (trial, cls_name, it)
            s_ref, z_ref = ob.calculate_qparams(ob.min_val, ob.max_val)
            s_or, z_or = st.qparams()
            assert np.array_equal(_bits(s_or), _bits(s_ref.numpy())), (trial, "scale")
            assert np.array_equal(np.asarray(z_or, dtype=np.float64).reshape(-1),
                                  z_ref.numpy().astype(np.float64).reshape(-1)), (trial, "zero_point")


def test_lsqplus_forward_backward_random_cases(ref):
    """util_quant.py:48-55 with autograd against the oracle's closed-form forward / backward."""
    _, U = ref
    from oracle import fake_quant_oracle as FQ
    gen = torch.Generator().manual_seed(424242)
    for trial in range(40):
        bit = int(torch.randint(3, 9, (1,), generator=gen))
        qmin, qmax = 0, (1 << bit) - 1
        shape = (int(torch.randint(1, 6, (1,), generator=gen)), int(torch.randint(1, 40, (1,), generator=gen)))
        x = (torch.randn(*shape, generator=gen) * 2).requires_grad_(True)
        scale = torch.empty(1).uniform_(0.01, 0.5, generator=gen).requires_grad_(True)
        zp = torch.empty(1).uniform_(qmin, qmax, generator=gen).requires_grad_(True)
        g = float(FQ.lsqplus_grad_factor(x.numel(), qmax))
        y = U.fake_quantize_learnableplus_per_tensor_affine_training(x, scale, zp, qmin, qmax, g)
        gy = torch.randn(*shape, generator=gen)
        y.backward(gy)
        _, y_o = FQ.fake_quantize_learnableplus_per_tensor(x.detach().numpy(), F32(scale.item()), F32(zp.item()), qmin, qmax, g)
        assert np.array_equal(_bits(y_o), _bits(y.detach().numpy())), (trial, "y")
        dx, ds, dz = FQ.lsqplus_backward_per_tensor(x.detach().numpy(), gy.numpy(), F32(scale.item()), F32(zp.item()), qmin, qmax, g)
        assert np.array_equal(_bits(dx), _bits(x.grad.numpy())), (trial, "dx")
        np.testing.assert_allclose(ds, scale.grad.item(), rtol=3e-5, atol=1e-7)
        np.testing.assert_allclose(dz, zp.grad.item(), rtol=3e-5, atol=1e-7)


def test_msefast_equals_reference_when_the_loss_is_summed_like_torch(ref):
    """MSEFast, per-channel and per-tensor (1-D and nested 2-D searches): with torch's own fp32 mean plugged into the
    oracle's loss -- the one step whose order is the machine's, not the algorithm's -- the oracle reproduces the
    reference's ranges bit for bit and spends the same number of loss evaluations.  Everything else (float64
    qparams, fp32 fake-quant, np.float32 function values inside scipy's bounded minimiser) is restated exactly."""
    O, _ = ref
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(77)
    old = OB.MEAN_LIKE_TORCH
    OB.MEAN_LIKE_TORCH = lambda sq: torch.from_numpy(np.ascontiguousarray(sq)).mean().numpy()
    try:
        for cols, bit in ((768, 4), (3072, 4), (96, 6)):
            w = torch.randn(24, cols, generator=gen) * 0.05
            ob = O.MSEFastObserver(bit=bit, symmetric=True, ch_axis=0)
            ob(w)
            st = OB.ObserverState(bit=bit, symmetric=True, ch_axis=0)
            OB.observe_msefast(st, w.numpy())
            assert np.array_equal(st.min_val, ob.min_val.numpy()) and np.array_equal(st.max_val, ob.max_val.numpy()), cols
        for sym, shape in ((True, (4, 16, 32)), (False, (4, 16, 32)), (False, (2, 8, 64))):
            x = torch.randn(*shape, generator=gen)
            x[..., 1] *= 9.0
            ob = O.AvgMSEFastObserver(bit=6, symmetric=sym, ch_axis=-1)
            st = OB.ObserverState(bit=6, symmetric=sym)
            for it in range(2):
                ob(x * (it + 1))
                OB.observe_msefast(st, x.numpy() * (it + 1), average=True)
                assert float(st.min_val) == float(ob.min_val) and float(st.max_val) == float(ob.max_val), (sym, shape, it)
        # one-sided data over four batches: the dtype of min_val decides whether the NEXT batch is searched on a float64
        # copy (observer.py:524 / 549) -- float32 for ever on non-negative data (the zeros_like of :491), float64 from the
        # second batch on otherwise; both observers, values and dtypes
        for cls, avg in ((O.MSEFastObserver, False), (O.AvgMSEFastObserver, True)):
            for side in ("pos", "neg"):
                ob = cls(bit=6, symmetric=False, ch_axis=-1)
                st = OB.ObserverState(bit=6, symmetric=False)
                for it in range(4):
                    x = torch.randn(3, 7, 16, generator=gen).abs() * (1.0 + 0.3 * it) + 1e-3
                    x = x if side == "pos" else -x
                    ob(x)
                    OB.observe_msefast(st, x.numpy(), average=avg)
                    assert float(st.min_val) == float(ob.min_val) and float(st.max_val) == float(ob.max_val), (cls.__name__, side, it)
                    assert str(np.asarray(st.min_val).dtype) == str(ob.min_val.dtype).replace("torch.", ""), (cls.__name__, side, it)
    finally:
        OB.MEAN_LIKE_TORCH = old


def test_shipped_quant_sections_equal_the_reference_yaml():
    """ptq.SHIPPED_QUANT_SECTIONS against the `quant:` section of every config.yaml under the reference's exp/ (15 files,
    four distinct sections): same keys and values."""
    import glob
    import yaml
    from outlier_suppression_amd import ptq
    files = sorted(glob.glob(os.path.join(REF, "exp", "**", "config.yaml"), recursive=True))
    assert len(files) >= 15

    def plain(ns):
        return {k: (plain(v) if hasattr(v, "__dict__") else v) for k, v in vars(ns).items()}
    seen = set()
    for f in files:
        quant = yaml.safe_load(open(f))["quant"]
        kind = os.path.relpath(f, os.path.join(REF, "exp")).split(os.sep)[1]
        assert kind in ptq.SHIPPED_QUANT_SECTIONS, f
        assert plain(ptq.SHIPPED_QUANT_SECTIONS[kind]) == quant, f
        assert plain(ptq.namespace(quant)) == quant
        seen.add(kind)
    assert seen == set(ptq.SHIPPED_QUANT_SECTIONS)


def test_qparams_of_mixed_dtype_statistics(ref):
    """calculate_qparams on the statistics per-tensor MSEFast leaves behind: a float32 min_val beside a float64 max_val (and
    the other three combinations) -- torch promotes to float64 as soon as either is; values and dtype equal the reference's."""
    O, _ = ref
    from oracle import observer_oracle as OB
    rng = np.random.default_rng(0)
    for sym in (False, True):
        ob = O.AvgMSEFastObserver(bit=8, symmetric=sym, ch_axis=-1)
        for k in range(2000):
            dts = [(np.float32, np.float64), (np.float64, np.float32), (np.float32, np.float32), (np.float64, np.float64)][k % 4]
            mn, mx = dts[0](-rng.random() * 2), dts[1](rng.random() * 3)
            s, z = ob.calculate_qparams(torch.tensor(mn), torch.tensor(mx))
            so, zo = OB.calculate_qparams(np.asarray(mn), np.asarray(mx), ob.quant_min, ob.quant_max, sym)
            assert s.item() == float(so) and float(z) == float(zo), (sym, k, dts)
            assert str(s.dtype).replace("torch.", "") == str(np.asarray(so).dtype), (sym, k, dts)
