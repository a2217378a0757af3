"""GPU parity: the HIP path (through the C ABI) against the golden vectors produced by the
reference and against the oracle on seeded inputs.  Run with ``-m gpu`` on an MI355X.

Bar (BASELINE.json north_star): x_quant bit-exact, dequantised output bit-exact (the
stated tolerance is 1e-5; we get equality), statistics (min/max/scale/zero_point) bit-exact;
float sums (LSQ+ dscale / dzero_point) to 2e-5 relative.
"""
from types import SimpleNamespace as NS

import numpy as np
from conftest import bits_equal
import pytest
import torch

pytestmark = pytest.mark.gpu

F32 = np.float32


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from outlier_suppression_amd import _hip
    _hip.load()          # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def N(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------------- fake-quant

def test_per_tensor_golden(golden, eq32, dev):
    from outlier_suppression_amd import ops
    g = golden("fake_quant")
    for k in range(int(g["n_per_tensor"])):
        scale, zp, qmin, qmax = g[f"pt{k}_meta"][:4]
        x = T(g[f"pt{k}_x"], dev)
        s = torch.tensor([scale], dtype=torch.float32, device=dev)
        for zdt in (torch.int32, torch.float32):
            z = torch.tensor([zp], device=dev).to(zdt)
            y, xq = ops.fake_quant_per_tensor(x, s, z, int(qmin), int(qmax), return_quantized=True)
            assert eq32(N(xq), g[f"pt{k}_xq"]), f"x_quant case {k}"
            assert eq32(N(y), g[f"pt{k}_y"]), f"dequant case {k}"
        # misaligned base pointer -> scalar kernel, odd length -> vector kernel tail
        y2 = ops.fake_quant_per_tensor(x[1:], s, z, int(qmin), int(qmax))
        assert eq32(N(y2), g[f"pt{k}_y"][1:])
        y3 = ops.fake_quant_per_tensor(x[: x.numel() - 3].clone(), s, z, int(qmin), int(qmax))
        assert eq32(N(y3), g[f"pt{k}_y"][:-3])


def test_per_tensor_layouts(eq32, dev):
    """Dense permuted views ([B,h,T,d] and its transpose, quant_bert.py:148-150) and a non-dense slice."""
    from outlier_suppression_amd import ops
    from oracle import fake_quant_oracle as FQ
    gen = torch.Generator().manual_seed(3)
    base = torch.randn(4, 16, 2, 8, generator=gen) * 3
    s = torch.tensor([0.11], device=dev)
    z = torch.tensor([29], dtype=torch.int32, device=dev)
    for view in (lambda t: t.permute(0, 2, 1, 3), lambda t: t.permute(0, 2, 1, 3).transpose(-1, -2),
                 lambda t: t[:, 0], lambda t: t[:, ::2], lambda t: t[..., 1:7]):
        xv = view(base.to(dev))
        y = ops.fake_quant_per_tensor(xv, s, z, 0, 63)
        assert y.shape == xv.shape
        _, want = FQ.fake_quantize_per_tensor_affine(view(base).numpy(), F32(0.11), 29, 0, 63)
        assert eq32(N(y), want)
    # head-split views come back in the layout the following batched matmul wants: contiguous, or (key) the
    # transpose of a contiguous tensor -- so torch.matmul does not have to copy them out first
    big = (torch.randn(8, 32, 12, 64, generator=gen) * 3).to(dev)
    q_view = big.permute(0, 2, 1, 3)                       # [B, h, T, d] through [B, T, h, d] memory
    yq = ops.fake_quant_per_tensor(q_view, s, z, 0, 63)
    assert yq.is_contiguous()
    yk = ops.fake_quant_per_tensor(q_view.transpose(-1, -2), s, z, 0, 63)
    assert yk.shape == (8, 12, 64, 32) and yk.transpose(-1, -2).is_contiguous()
    _, want = FQ.fake_quantize_per_tensor_affine(q_view.cpu().numpy(), F32(0.11), 29, 0, 63)
    assert eq32(N(yq), want) and eq32(N(yk.transpose(-1, -2)), want)
    # under autograd the gradient still comes back in x's own layout
    xg = q_view.clone().requires_grad_(True)
    sp = torch.nn.Parameter(torch.tensor([0.11], device=dev))
    zp = torch.nn.Parameter(torch.tensor([29.0], device=dev))
    out = ops.fake_quant(xg.transpose(-1, -2), sp, zp, -1, 0, 63, ops.PARAM_LSQPLUS, 1e-3)
    out.backward(torch.ones_like(out))
    assert xg.grad.shape == xg.shape and torch.isfinite(xg.grad).all() and sp.grad is not None


def test_per_channel_golden(golden, eq32, dev):
    from outlier_suppression_amd import ops
    g = golden("fake_quant")
    for k in range(int(g["n_per_channel"])):
        ch_axis, qmin, qmax, bit, sym = (int(v) for v in g[f"pc{k}_meta"])
        x = T(g[f"pc{k}_x"], dev)
        y, xq = ops.fake_quant_per_channel(x, T(g[f"pc{k}_scale"], dev), T(g[f"pc{k}_zp"], dev), ch_axis, qmin, qmax,
                                           return_quantized=True)
        assert eq32(N(xq), g[f"pc{k}_xq"]) and eq32(N(y), g[f"pc{k}_y"])


def test_per_channel_wide_rows(eq32, dev):
    """Row kernel (inner >= 64, % 4 == 0): weights [C_out, C_in] with ch_axis = 0, against the oracle."""
    from outlier_suppression_amd import ops
    from oracle import fake_quant_oracle as FQ, observer_oracle as OB
    gen = torch.Generator().manual_seed(4)
    for shape in ((37, 768), (5, 3072), (130, 64), (3, 68)):
        w = torch.randn(*shape, generator=gen) * 0.05
        st = OB.ObserverState(bit=6, symmetric=True, ch_axis=0)
        OB.observe_minmax(st, w.numpy())
        scale, zp = st.qparams()
        y, xq = ops.fake_quant_per_channel(w.to(dev), T(scale, dev), T(zp, dev), 0, -32, 31, return_quantized=True)
        wq, wy = FQ.fake_quantize_per_channel_affine(w.numpy(), scale, zp, 0, -32, 31)
        assert eq32(N(xq), wq) and eq32(N(y), wy)


def test_lsqplus_golden(golden, eq32, dev):
    from outlier_suppression_amd.quantization import util_quant as U
    g = golden("lsqplus")
    for k in range(int(g["n"])):
        scale, zp, qmin, qmax, gf = g[f"c{k}_meta"]
        x = T(g[f"c{k}_x"], dev).requires_grad_(True)
        s = torch.tensor([scale], dtype=torch.float32, device=dev, requires_grad=True)
        z = torch.tensor([zp], dtype=torch.float32, device=dev, requires_grad=True)
        y = U.fake_quantize_learnableplus_per_tensor_affine_training(x, s, z, int(qmin), int(qmax), gf)
        y.backward(T(g[f"c{k}_gy"], dev))
        assert eq32(N(y), g[f"c{k}_y"])
        assert eq32(N(x.grad), g[f"c{k}_dx"])
        np.testing.assert_allclose(N(s.grad), g[f"c{k}_ds"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(N(z.grad), g[f"c{k}_dzp"], rtol=2e-5, atol=1e-7)
    qmin, qmax, gf = int(g["pc_meta"][1]), int(g["pc_meta"][2]), g["pc_meta"][3]
    x = T(g["pc_x"], dev).requires_grad_(True)
    s = T(g["pc_scale"], dev).requires_grad_(True)
    z = T(g["pc_zp"], dev).requires_grad_(True)
    y = U.fake_quantize_learnableplus_per_channel_affine_training(x, s, z, 0, qmin, qmax, gf)
    y.backward(T(g["pc_gy"], dev))
    assert eq32(N(y), g["pc_y"]) and eq32(N(x.grad), g["pc_dx"])
    np.testing.assert_allclose(N(s.grad), g["pc_ds"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(N(z.grad), g["pc_dzp"], rtol=2e-5, atol=1e-7)


def test_lsqplus_gradients_equal_reference_in_its_summation_order(golden, eq32, dev):
    """osq_set_tuning("bwd_sum_order", 8): the backward adds its four reductions in fp32 in the order torch's CPU kernel
    does (lsq_bwd_tensor_aten_kernel) -- scale.grad and zero_point.grad then equal the reference's own autograd run
    (tests/golden/lsqplus.npz) BIT FOR BIT, like y and dx always do."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import util_quant as U
    g = golden("lsqplus")
    ops.set_tuning("bwd_sum_order", 8)
    try:
        for k in range(int(g["n"])):
            scale, zp, qmin, qmax, gf = g[f"c{k}_meta"]
            x = T(g[f"c{k}_x"], dev).requires_grad_(True)
            s = torch.tensor([scale], dtype=torch.float32, device=dev, requires_grad=True)
            z = torch.tensor([zp], dtype=torch.float32, device=dev, requires_grad=True)
            y = U.fake_quantize_learnableplus_per_tensor_affine_training(x, s, z, int(qmin), int(qmax), gf)
            y.backward(T(g[f"c{k}_gy"], dev))
            assert eq32(N(y), g[f"c{k}_y"]) and eq32(N(x.grad), g[f"c{k}_dx"])
            assert eq32(N(s.grad), g[f"c{k}_ds"]) and eq32(N(z.grad), g[f"c{k}_dzp"]), (k, N(s.grad), g[f"c{k}_ds"], N(z.grad), g[f"c{k}_dzp"])
    finally:
        ops.set_tuning("bwd_sum_order", 0)      # the default tier (order-free backward sums since round 5)


def test_lsqplus_per_channel_gradients_equal_reference_in_its_summation_order(golden, eq32, dev):
    """The per-channel learnable quantizer (weights, ch_axis = 0) under the strict switch: every row's four reductions in
    torch's order -- scale.grad and zero_point.grad of tests/golden/lsqplus.npz's per-channel case bit for bit."""
    import outlier_suppression_amd as osq
    from outlier_suppression_amd.quantization import util_quant as U
    g = golden("lsqplus")
    qmin, qmax, gf = int(g["pc_meta"][1]), int(g["pc_meta"][2]), g["pc_meta"][3]
    osq.set_strict(True)
    try:
        x = T(g["pc_x"], dev).requires_grad_(True)
        s = T(g["pc_scale"], dev).requires_grad_(True)
        z = T(g["pc_zp"], dev).requires_grad_(True)
        y = U.fake_quantize_learnableplus_per_channel_affine_training(x, s, z, 0, qmin, qmax, gf)
        y.backward(T(g["pc_gy"], dev))
        assert eq32(N(y), g["pc_y"]) and eq32(N(x.grad), g["pc_dx"])
        assert eq32(N(s.grad), g["pc_ds"]) and eq32(N(z.grad), g["pc_dzp"]), (N(s.grad), g["pc_ds"], N(z.grad), g["pc_dzp"])
    finally:
        osq.reset_tier()


def test_lsq_backward_determinism_and_size(dev):
    """Grid-wide reduction: same bits run to run, and correct at a size with many workgroups."""
    from outlier_suppression_amd import ops
    from oracle import fake_quant_oracle as FQ
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(64, 128, 768, generator=gen)
    gy = torch.randn(64, 128, 768, generator=gen)
    s = torch.tensor([0.07], device=dev)
    z = torch.tensor([31.4], device=dev)
    gf = FQ.lsqplus_grad_factor(x.numel(), 63)
    outs = [ops.lsq_backward_per_tensor(x.to(dev), gy.to(dev), s, z, 0, 63, ops.PARAM_LSQPLUS, gf) for _ in range(3)]
    for dx, ds, dz in outs[1:]:
        assert torch.equal(dx, outs[0][0]) and torch.equal(ds, outs[0][1]) and torch.equal(dz, outs[0][2])
    dx_o, ds_o, dz_o = FQ.lsqplus_backward_per_tensor(x.numpy(), gy.numpy(), F32(0.07), F32(31.4), 0, 63, gf)
    assert bits_equal(N(outs[0][0]), dx_o)
    np.testing.assert_allclose(N(outs[0][1])[0], ds_o, rtol=2e-5)
    np.testing.assert_allclose(N(outs[0][2])[0], dz_o, rtol=2e-5)


# ----------------------------------------------------------------------------------- observers

def test_calculate_qparams_golden(golden, eq32, dev):
    from outlier_suppression_amd import ops
    g = golden("qparams")
    for k in range(int(g["n"])):
        bit, sym, qmin, qmax = (int(v) for v in g[f"c{k}_meta"])
        scale, zp = ops.calculate_qparams(T(g["min"], dev), T(g["max"], dev), qmin, qmax, bool(sym))
        assert eq32(N(scale), g[f"c{k}_scale"])
        assert eq32(N(zp).astype(F32), g[f"c{k}_zp"].astype(F32))
        assert zp.dtype == (torch.int32 if sym else torch.float32)


def _as_layout(x, lay, dev):
    """Rebuild the strided view the reference model hands to the quantizer (same logical values)."""
    t = torch.from_numpy(x).to(dev)
    if lay in ("bhtd",):        # [B,h,T,d] view of [B,T,h,d] memory (transpose_for_scores, quant_bert.py:128-132)
        return t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    if lay == "bhdt":           # key_layer.transpose(-1,-2): [B,h,d,T] view of [B,T,h,d] memory
        return t.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
    return t


def test_observer_sequences_golden(golden, eq32, dev):
    from outlier_suppression_amd.quantization.quantized_module import ObserverDict
    g = golden("observers")
    for k in range(int(g["n"])):
        obs_name, lay, seq_pos, masked, name, p = (str(v) for v in g[f"c{k}_info"])
        seq_pos, masked = int(seq_pos), bool(int(masked))
        for strided in ((False, True) if lay in ("bhtd", "bhdt") else (False,)):
            ob = ObserverDict[obs_name](bit=6, symmetric=False, ch_axis=-1).to(dev)
            ob.set_name(name)
            if p:
                ob.set_percentile(float(p))
            xs, lens = g[f"c{k}_x"], g[f"c{k}_len"]
            for it in range(xs.shape[0]):
                x = _as_layout(xs[it], lay, dev) if strided else T(xs[it], dev)
                sp = seq_pos if (masked or obs_name == "AvgPruneMinMaxObserver") else -1
                ob(x, observation_mask=T(lens[it], dev) if masked else None, seq_pos=sp)
                assert eq32(N(ob.min_val), g[f"c{k}_min"][it]), (k, obs_name, lay, it, strided, "min")
                assert eq32(N(ob.max_val), g[f"c{k}_max"][it]), (k, obs_name, lay, it, strided, "max")
            scale, zp = ob.calculate_qparams(ob.min_val, ob.max_val)
            assert eq32(N(scale), g[f"c{k}_scale"]) and eq32(N(zp).astype(F32), g[f"c{k}_zp"].astype(F32))


def test_observer_midsize_golden(golden, dev):
    from outlier_suppression_amd import ops
    g = golden("observer_midsize")
    L = g["lengths"]
    B, Tn = len(L), 64
    tmin = np.full((B, Tn), 123.0, F32)   # padded slots hold junk the finaliser must ignore
    tmax = np.full((B, Tn), -321.0, F32)
    off = 0
    for b, n in enumerate(L):
        tmin[b, :n] = g["token_min"][off:off + n]
        tmax[b, :n] = g["token_max"][off:off + n]
        off += n
    cur = torch.empty(2, device=dev)
    for p, mn, mx in zip(g["percentiles"], g["mins"], g["maxs"]):
        ops.token_range_finalize(T(tmin.reshape(-1), dev), T(tmax.reshape(-1), dev), B, Tn, T(L, dev), True, float(p),
                                 ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur)
        assert N(cur)[0] == mn and N(cur)[1] == mx, (p, N(cur), mn, mx)


@pytest.mark.parametrize("shape,seq_pos", [((32, 128, 768), 1), ((8, 12, 128, 64), 2), ((8, 12, 64, 128), 3),
                                           ((4, 12, 128, 128), 2), ((16, 128, 3072), 1), ((8, 384, 768), 1),
                                           ((3, 5, 7, 6), 2), ((2, 9, 10), 1)])
def test_token_observer_vs_oracle(shape, seq_pos, eq32, dev):
    """Seeded BERT-base-shaped sites (SURVEY 8a size table) against the oracle, three batches each."""
    from outlier_suppression_amd.quantization.observer import AvgPruneMinMaxObserver, AvgMinMaxObserver
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(hash(shape) % 1000)
    Tn = shape[seq_pos]
    for cls, fn, p in ((AvgPruneMinMaxObserver, OB.observe_avg_prune_minmax, 0.93), (AvgMinMaxObserver, OB.observe_avg_minmax, None)):
        ob = cls(bit=6, symmetric=False).to(dev)
        ob.set_name("layer.x_post_act_fake_quantize.observer")
        st = OB.ObserverState(bit=6, symmetric=False, name=ob.name)
        if p is not None:
            ob.set_percentile(p)
            st.percentile = p
        for it in range(3):
            x = torch.randn(*shape, generator=gen)
            x.select(-1, 1).mul_(15.0)
            L = torch.randint(0 if it == 2 else 1, Tn + 1, (shape[0],), generator=gen)
            L[0] = Tn
            ob(x.to(dev), L.to(dev), seq_pos)
            fn(st, x.numpy(), L.numpy(), seq_pos)
            assert eq32(N(ob.min_val), st.min_val) and eq32(N(ob.max_val), st.max_val), (cls.__name__, it)
            assert ob.cnt == st.cnt


@pytest.mark.parametrize("shape,seq_pos,n_mask,name", [
    ((8, 12, 384, 384), 2, 8, "attention_probs"),          # configs[2]: SQuAD, T = 384 (tokens-first rows of 12 x 384 = 4608 features)
    ((8, 12, 384, 64), 2, 8, "query"),
    ((8, 384, 3072), 1, 8, "intermediate"),
    ((4, 1024, 1024), 1, 4, "final_layer_norm"),            # configs[4]: BART-large encoder, d = 1024, T = 1024
    ((4, 1024, 4096), 1, 4, "fc1"),
    ((16, 1024, 1024), 1, 4, "attention_probs"),            # 3-D probabilities [B*h, T, S] with a length-B mask: zip truncation
    ((64, 62, 1024), 1, 4, "attention_probs"),              # decoder cross-attention
    ((4, 62, 1024), 1, 4, "decoder_ln"),
])
def test_config_site_shapes_vs_oracle(shape, seq_pos, n_mask, name, eq32, dev):
    """The site shapes of BASELINE configs 2 (SQuAD, T = 384) and 4 (BART-large, d = 1024, T = 1024) at their real
    sizes against the oracle (SURVEY 8a size table; quirk 9: a 3-D probability tensor with a length-B mask is cut to
    its first B rows by remove_padding's zip)."""
    from outlier_suppression_amd.quantization.observer import AvgPruneMinMaxObserver
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(hash(shape) % 1000)
    Tn = shape[seq_pos]
    ob = AvgPruneMinMaxObserver(bit=6, symmetric=False).to(dev)
    ob.set_name(f"layer.{name}_post_act_fake_quantize.observer")
    st = OB.ObserverState(bit=6, symmetric=False, name=ob.name)
    ob.set_percentile(0.93)
    st.percentile = 0.93
    for it in range(2):
        x = torch.rand(*shape, generator=gen) if name == "attention_probs" else torch.randn(*shape, generator=gen)
        if name != "attention_probs":
            x.select(-1, 1).mul_(15.0)
        L = torch.randint(1, Tn + 1, (n_mask,), generator=gen)
        L[0] = Tn
        ob(x.to(dev), L.to(dev), seq_pos)
        OB.observe_avg_prune_minmax(st, x.numpy(), L.numpy(), seq_pos)
        assert eq32(N(ob.min_val), st.min_val) and eq32(N(ob.max_val), st.max_val), (shape, it, N(ob.min_val), st.min_val, N(ob.max_val), st.max_val)
    scale, zp = ob.calculate_qparams(ob.min_val, ob.max_val)
    s_o, z_o = st.qparams()
    assert eq32(N(scale), s_o) and bits_equal(N(zp), z_o)


def test_flat_and_channel_observers_vs_oracle(eq32, dev):
    from outlier_suppression_amd.quantization.observer import MinMaxObserver, AvgMinMaxObserver
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(21)
    # flat, several sizes around vector/tail/grid boundaries
    for n in (1, 3, 4, 5, 1023, 4096, 1 << 20, (1 << 22) + 7):
        ob = AvgMinMaxObserver(bit=6).to(dev)
        st = OB.ObserverState(bit=6)
        for it in range(2):
            x = torch.randn(n, generator=gen)
            ob(x.to(dev))
            OB.observe_avg_minmax(st, x.numpy())
        assert eq32(N(ob.min_val), st.min_val) and eq32(N(ob.max_val), st.max_val), n
    # per-channel weights, ch_axis 0 (row kernel) and a middle axis (generic kernel)
    for shape, ax in (((768, 768), 0), ((77, 3072), 0), ((5, 12), 0), ((4, 6, 10), 1), ((6, 4, 3, 3), 0)):
        ob = MinMaxObserver(bit=6, symmetric=True, ch_axis=ax).to(dev)
        st = OB.ObserverState(bit=6, symmetric=True, ch_axis=ax)
        for it in range(2):
            w = torch.randn(*shape, generator=gen) * 0.05
            ob(w.to(dev))
            OB.observe_minmax(st, w.numpy())
        assert eq32(N(ob.min_val), st.min_val) and eq32(N(ob.max_val), st.max_val), shape
        scale, zp = ob.calculate_qparams(ob.min_val, ob.max_val)
        s_o, z_o = st.qparams()
        assert eq32(N(scale), s_o) and bits_equal(N(zp), z_o)


def test_nan_poisons_statistics(dev):
    from outlier_suppression_amd.quantization.observer import MinMaxObserver
    x = torch.randn(4, 8, 16)
    x[1, 2, 3] = float("nan")
    ob = MinMaxObserver(bit=6).to(dev)
    ob(x.to(dev))
    assert torch.isnan(ob.min_val).item() and torch.isnan(ob.max_val).item()
    ob2 = MinMaxObserver(bit=6).to(dev)
    ob2(x.to(dev), torch.tensor([8, 1, 8, 8], device=dev), 1)       # the NaN token is padding -> ignored
    assert not torch.isnan(ob2.min_val).item()


# ----------------------------------------------------------------------------------- modules

def test_module_traces_golden(golden, eq32, dev):
    from outlier_suppression_amd.quantization import Quantizer
    g = golden("modules")
    for k in range(int(g["n"])):
        quantizer, observer, bit, sym, ch_axis, kind, sdt, zdt = (str(v) for v in g[f"c{k}_info"])
        cfg = NS(quantizer=quantizer, observer=observer, bit=int(bit), symmetric=bool(int(sym)), ch_axis=int(ch_axis))
        q = Quantizer(None, cfg).to(dev)
        q.observer.set_name("m.x_post_act_fake_quantize.observer")
        q.observer.set_percentile(0.9)
        q.enable_observer()
        q.disable_fake_quant()
        xs, lens = g[f"c{k}_x"], g[f"c{k}_len"]
        for it in range(xs.shape[0]):
            x = T(xs[it], dev)
            r = q(x, observation_mask=T(lens[it], dev), seq_pos=1) if kind == "act" else q(x)
            assert r is x
            assert eq32(N(q.scale).reshape(-1), g[f"c{k}_scale"][it].reshape(-1)), (k, it)
            assert eq32(N(q.zero_point).astype(F32).reshape(-1), g[f"c{k}_zp"][it].astype(F32).reshape(-1)), (k, it)
        assert str(q.scale.dtype) == sdt and str(q.zero_point.dtype) == zdt
        assert sorted(q.state_dict().keys()) == [str(s) for s in g[f"c{k}_sdkeys"]]
        q.disable_observer()
        q.enable_fake_quant()
        xt = T(xs[-1], dev).requires_grad_(True)
        y = q(xt, observation_mask=T(lens[-1], dev), seq_pos=1) if kind == "act" else q(xt)
        y.backward(T(g[f"c{k}_gy"], dev))
        assert eq32(N(y), g[f"c{k}_y"]), (k, quantizer)
        assert eq32(N(xt.grad), g[f"c{k}_dx"]), (k, quantizer)
        # float sums: the reference accumulates in fp32 (its own rounding is ~1e-6 absolute on
        # per-channel sums that cancel), the kernel in float64
        if f"c{k}_ds" in g.files:
            np.testing.assert_allclose(N(q.scale.grad), g[f"c{k}_ds"], rtol=2e-5, atol=2e-6)
        if f"c{k}_dzp" in g.files:
            np.testing.assert_allclose(N(q.zero_point.grad), g[f"c{k}_dzp"], rtol=2e-5, atol=2e-6)


def test_learnable_sanitize(dev):
    """fake_quant.py:188-191: |scale|, floor eps, clamp zero_point -- while the observer is off."""
    from outlier_suppression_amd.quantization import Quantizer
    cfg = NS(quantizer="LSQPlusFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    q = Quantizer(None, cfg).to(dev)
    q.scale.data.fill_(-0.25)
    q.zero_point.data.fill_(99.0)
    q(torch.randn(4, 4, device=dev))
    assert q.scale.item() == 0.25 and q.zero_point.item() == 63.0
    q.scale.data.fill_(0.0)
    q.zero_point.data.fill_(-3.0)
    q(torch.randn(4, 4, device=dev))
    assert q.scale.item() == pytest.approx(float(torch.finfo(torch.float32).eps)) and q.zero_point.item() == 0.0
    # with fake-quant on, the repair is folded into the fake-quant launch: same parameters afterwards and
    # the output of the separate repair + quantise sequence
    from outlier_suppression_amd import ops
    x = torch.randn(6, 40, device=dev) * 3
    for cls, zp0 in (("LSQPlusFakeQuantize", 99.0), ("LSQPlusFakeQuantize", -7.5), ("LSQFakeQuantize", 5)):
        cfg = NS(quantizer=cls, observer="MinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
        q = Quantizer(None, cfg).to(dev)
        q.enable_fake_quant()
        q.scale.data.fill_(-0.25)
        q.zero_point.data.fill_(zp0)
        y = q(x)
        zp_want = min(max(zp0, 0.0), 63.0) if cls == "LSQPlusFakeQuantize" else zp0      # LSQ leaves its int zero_point alone
        assert q.scale.item() == 0.25 and q.zero_point.item() == zp_want
        s_ref = torch.tensor([0.25], device=dev)
        z_ref = torch.tensor([zp_want], device=dev, dtype=q.zero_point.dtype)
        y_ref = ops.fake_quant_per_tensor(x, s_ref, z_ref, 0, 63, q.param_mode, q._grad_factor(x))
        assert torch.equal(y, y_ref), cls
        # and under autograd (learn_scale): gradients as with already-clean parameters
        q.scale.data.fill_(-0.25)
        q.zero_point.data.fill_(zp0)
        xg = x.clone().requires_grad_(True)
        q(xg).sum().backward()
        assert q.scale.item() == 0.25 and q.scale.grad is not None and torch.isfinite(q.scale.grad).all()
    # per-channel learnable quantizers keep the separate repair launch
    cfg = NS(quantizer="LSQPlusFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=False, ch_axis=0)
    q = Quantizer(None, cfg).to(dev)
    q.enable_fake_quant()
    q.scale = torch.nn.Parameter(torch.full((6,), -0.5, device=dev))
    q.zero_point = torch.nn.Parameter(torch.full((6,), 70.0, device=dev))
    q(x)
    assert (q.scale.data == 0.5).all() and (q.zero_point.data == 63.0).all()


def test_gamma_golden(golden, eq32, dev):
    from outlier_suppression_amd import ops
    g = golden("gamma")
    W = T(g["W"], dev)
    ops.gamma_fold_(W, T(g["gamma"], dev))
    assert eq32(N(W), g["W_folded"])
    assert eq32(N(ops.gamma_split_bias(T(g["beta"], dev), T(g["gamma"], dev))), g["split_bias"])
    assert eq32(N(ops.gamma_residual(T(g["x"], dev), T(g["hidden"], dev))), g["res_before"])
    assert eq32(N(ops.gamma_residual(T(g["x"], dev), T(g["hidden"], dev), T(g["gamma"], dev))), g["res_after"])


def test_fused_residual_layernorm_fake_quant(dev):
    """One-launch LayerNorm site (residual + LayerNorm + affine / beta-over-gamma shift + fake-quant) against the
    eager sequence it replaces (util_layernorm.py:14-18, 32-37, 49-52).  The normalisation cannot be bit-compared
    with torch's kernel (different summation), so: un-quantised output within 2e-6 absolute of eager; quantised
    output equal to fake_quant(own normalised output) bit for bit, and equal to the eager chain except where the
    normalised value sits within 4e-6 of a rounding boundary (there it may differ by exactly one step)."""
    import torch.nn.functional as F
    from outlier_suppression_amd import ops
    gen = torch.Generator().manual_seed(31)
    for shape in ((32, 128, 768), (4, 7, 64), (3, 5, 4096), (6, 1024), (2, 3, 260)):
        H = shape[-1]
        x = (torch.randn(*shape, generator=gen) * 2).to(dev)
        hid = torch.randn(*shape, generator=gen).to(dev)
        gamma = (torch.rand(H, generator=gen) + 0.5).to(dev)
        w = (torch.rand(H, generator=gen) + 0.5).to(dev)
        b = torch.randn(H, generator=gen).to(dev)
        scale = torch.tensor([0.11], device=dev)
        zp = torch.tensor([29.0], device=dev)
        for use_hid, use_gamma, use_w, use_b, eps in ((1, 1, 0, 1, 1e-5), (1, 0, 1, 1, 1e-12), (0, 0, 1, 1, 1e-12), (0, 0, 0, 0, 1e-5)):
            r = x
            if use_hid:
                r = (x * gamma if use_gamma else x) + hid
            ref = F.layer_norm(r, (H,), None, None, eps)
            if use_w:
                ref = ref * w
            if use_b:
                ref = ref + b
            args = (x, hid if use_hid else None, gamma if (use_hid and use_gamma) else None, w if use_w else None,
                    b if use_b else None, eps)
            y = ops.residual_layernorm_fake_quant(*args)
            assert (y - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item()), (shape, use_hid, use_w)
            yq = ops.residual_layernorm_fake_quant(*args, quant=(scale, zp, 0, 63, ops.PARAM_LSQPLUS, 1e-4))
            assert torch.equal(yq, ops.fake_quant_per_tensor(y, scale, zp, 0, 63, ops.PARAM_LSQPLUS, 1e-4))
            eager = ops.fake_quant_per_tensor(ref.contiguous(), scale, zp, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
            diff = yq != eager
            if diff.any():
                frac = ref / scale.item()
                near_tie = ((frac - frac.floor()) - 0.5).abs() < 4e-6 / scale.item()
                assert bool((near_tie | ~diff).all()), shape
                assert ((yq - eager).abs()[diff] - scale.item()).abs().max().item() < 1e-5
                assert diff.float().mean().item() < 1e-3
    # layouts the kernel rejects surface as an error, and the modules fall back to the eager sequence
    with pytest.raises(RuntimeError):
        ops.residual_layernorm_fake_quant(torch.randn(4, 102, device=dev), None, None, None, None, 1e-5)
    # module level: fused forward == eager forward of the same modules, incl. the LSQ+ parameter repair
    from outlier_suppression_amd import util_layernorm as UL
    cfg = NS(quantizer="LSQPlusFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    ln = torch.nn.LayerNorm(64, eps=1e-12)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(64, generator=gen) + 0.5)
        ln.bias.copy_(torch.randn(64, generator=gen))
    ln = ln.to(dev)
    for cls in (UL.QuantizedLayerNorm, UL.QuantizedSplitLayerNorm):
        mod = cls(ln, None, cfg).to(dev)
        res = UL.GammaResidual()
        res.set_gamma(ln.weight)
        res = res.to(dev)
        q = mod.layernorm_post_act_fake_quantize
        xs, hs = torch.randn(5, 9, 64, generator=gen).to(dev), torch.randn(5, 9, 64, generator=gen).to(dev)
        L = torch.tensor([9, 3, 1, 9, 5], device=dev)
        outs = {}
        fuse_default = UL.FUSE_LAYERNORM
        for fuse in (True, False):
            UL.FUSE_LAYERNORM = fuse
            try:
                with torch.no_grad():
                    q.enable_observer(); q.disable_fake_quant()
                    q.observer.min_val.fill_(float("inf")); q.observer.max_val.fill_(float("-inf")); q.observer.cnt = 0
                    y_obs = UL.residual_layernorm(res, mod, xs, hs, L)
                    q.disable_observer(); q.enable_fake_quant()
                    q.scale.data.neg_()                               # the repair must run in both forms
                    y_q = UL.residual_layernorm(res, mod, xs, hs, L)
                    outs[fuse] = (y_obs.clone(), y_q.clone(), q.scale.item(), q.zero_point.item())
            finally:
                UL.FUSE_LAYERNORM = fuse_default
        assert (outs[True][0] - outs[False][0]).abs().max().item() < 4e-6 * max(1.0, outs[False][0].abs().max().item())
        assert outs[True][2] > 0 and abs(outs[True][2] - outs[False][2]) < 1e-6 * outs[False][2]
        assert (outs[True][1] - outs[False][1]).abs().max().item() <= outs[False][2] * 1.001   # at most one step, at ties
        # the observed range (hence scale) may differ by an ulp between the two forms: count whole-step flips only
        flips = (outs[True][1] - outs[False][1]).abs() > 0.5 * outs[False][2]
        assert flips.float().mean().item() < 5e-3


def test_quantized_operators_forward(eq32, dev):
    """QLinear / QConv2d / QEmbedding (quantized_module.py:39-100): weight observed per output channel on the
    first call, fake-quantised on every call, then the stock functional op -- against the oracle's weights."""
    import torch.nn.functional as F
    from outlier_suppression_amd.quantization import Quantizer
    from oracle import observer_oracle as OB, fake_quant_oracle as FQ
    torch.manual_seed(3)
    cfg = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    mods = [(torch.nn.Linear(40, 12), torch.randn(5, 40)),
            (torch.nn.Conv2d(3, 6, 3, padding=1), torch.randn(2, 3, 8, 8)),
            (torch.nn.Embedding(50, 16, padding_idx=0), torch.randint(0, 50, (4, 7)))]
    for fp_mod, inp in mods:
        qm = Quantizer(fp_mod, cfg).to(dev)
        wq = qm.weight_fake_quant
        wq.enable_observer()
        wq.enable_fake_quant()
        with torch.no_grad():
            out = qm(inp.to(dev))
        W = fp_mod.weight.detach().numpy()
        st = OB.ObserverState(bit=6, symmetric=True, ch_axis=0)
        OB.observe_minmax(st, W)
        scale, zp = st.qparams()
        assert eq32(N(wq.scale), scale) and bits_equal(N(wq.zero_point), zp)
        _, Wq = FQ.fake_quantize_per_channel_affine(W, scale, zp, 0, -32, 31)
        Wq_t = torch.from_numpy(Wq).to(dev)
        with torch.no_grad():
            if isinstance(fp_mod, torch.nn.Linear):
                ref = F.linear(inp.to(dev), Wq_t, fp_mod.bias.to(dev))
            elif isinstance(fp_mod, torch.nn.Conv2d):
                ref = F.conv2d(inp.to(dev), Wq_t, fp_mod.bias.to(dev), padding=1)
            else:
                ref = F.embedding(inp.to(dev), Wq_t, padding_idx=0)
        assert torch.equal(out, ref), type(fp_mod).__name__


def test_gelu_fake_quant_fused(dev):
    """One launch for GELU + fake-quant must equal F.gelu followed by the fake-quant launch bit for bit (the GELU is
    torch's own erf form in the same operation order), for Fixed and LSQ+ parameters, sizes with a scalar tail."""
    import torch.nn.functional as F
    from outlier_suppression_amd import ops, util_layernorm as UL
    from outlier_suppression_amd.quantization import Quantizer
    gen = torch.Generator().manual_seed(17)
    s = torch.tensor([0.037], device=dev)
    for shape in ((32, 128, 3072), (7, 13), (5,), (4, 1024, 4096)[:2]):
        x = (torch.randn(*shape, generator=gen) * 3).to(dev)
        x.view(-1)[:3] = torch.tensor([0.0, -0.0, 40.0], device=dev)[: min(3, x.numel())]
        g = F.gelu(x)
        for zp, mode, gf in ((torch.tensor([17], dtype=torch.int32, device=dev), ops.PARAM_FIXED, 1.0),
                             (torch.tensor([16.6], device=dev), ops.PARAM_LSQPLUS, 3e-4)):
            want = ops.fake_quant_per_tensor(g, s, zp, 0, 63, mode, gf)
            got = ops.gelu_fake_quant_per_tensor(x, s, zp, 0, 63, mode, gf)
            assert torch.equal(got, want), (shape, mode)
    # module-level helper: fused == two-step, also the LSQ+ parameter repair; falls back for other activations
    from transformers.activations import ACT2FN
    cfg = NS(quantizer="LSQPlusFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    q = Quantizer(None, cfg).to(dev)
    q.enable_fake_quant()
    x = (torch.randn(4, 9, 64, generator=gen) * 2).to(dev)
    outs = []
    for fuse in (True, False):
        UL.FUSE_ACTIVATION = fuse
        try:
            q.scale.data.fill_(-0.05)
            q.zero_point.data.fill_(80.0)
            with torch.no_grad():
                outs.append((UL.activation_fake_quant(ACT2FN["gelu"], q, x, None).clone(), q.scale.item(), q.zero_point.item()))
        finally:
            UL.FUSE_ACTIVATION = True
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:] and abs(outs[0][1] - 0.05) < 1e-9 and outs[0][2] == 63.0
    with torch.no_grad():
        r = UL.activation_fake_quant(torch.relu, q, x, None)
    assert torch.equal(r, q(torch.relu(x)))


# ----------------------------------------------------------------------------------- full size, properties

def test_full_size_properties(dev):
    """BASELINE's [256,128,768] activation: size-independent properties instead of an oracle pass."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import AvgPruneMinMaxObserver
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(256, 128, 768, device=dev, generator=gen)
    x[..., [7, 300, 511]] *= 20
    L = torch.randint(8, 129, (256,), device=dev, generator=gen)
    ob = AvgPruneMinMaxObserver(bit=6).to(dev)
    ob.set_name("x_post_act_fake_quantize.observer")
    ob.set_percentile(0.95)
    ob(x, L, 1)
    # cross-check with stock torch reductions on the device (min/max/sort are exact operations)
    valid = torch.arange(128, device=dev)[None, :] < L[:, None]
    tok_max = x.amax(-1)[valid]
    tok_min = x.amin(-1)[valid]
    up = torch.quantile(tok_max.abs().cpu(), 0.95).to(dev)
    lo = -torch.quantile(tok_min.abs().cpu(), 0.95).to(dev)
    assert ob.max_val.item() == tok_max[tok_max <= up].max().item()
    assert ob.min_val.item() == tok_min[tok_min >= lo].min().item()
    scale, zp = ob.calculate_qparams(ob.min_val, ob.max_val)
    y, xq = ops.fake_quant_per_tensor(x, scale.reshape(1), zp.reshape(1), 0, 63, return_quantized=True)
    assert torch.equal(xq, xq.round()) and xq.min().item() >= 0 and xq.max().item() <= 63
    # idempotence: quantising a dequantised tensor changes nothing
    y2 = ops.fake_quant_per_tensor(y, scale.reshape(1), zp.reshape(1), 0, 63)
    assert torch.equal(y, y2)
    # same arithmetic spelled with stock torch ops on a slice that is small enough for the CPU
    xs = x[:4].cpu()
    ref = (torch.clamp((xs / scale.item()).round() + zp.item(), 0, 63) - zp.item()) * scale.item()
    assert torch.equal(y[:4].cpu(), ref)


def test_empty_and_errors(dev):
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import MinMaxObserver
    s = torch.tensor([1.0], device=dev)
    z = torch.tensor([0], dtype=torch.int32, device=dev)
    y = ops.fake_quant_per_tensor(torch.empty(0, device=dev), s, z, 0, 63)
    assert y.numel() == 0
    ob = MinMaxObserver().to(dev)
    e = torch.empty(0, 4, device=dev)
    assert ob(e) is e and torch.isinf(ob.min_val).item()
    with pytest.raises(RuntimeError):
        ops.fake_quant_per_tensor(torch.randn(4), s, z, 0, 63)       # CPU tensor: no fallback
    with pytest.raises(TypeError):
        ops.fake_quant_per_tensor(torch.randn(4, device=dev).half(), s, z, 0, 63)


# ----------------------------------------------------------------------------------- MSEFast

def test_msefast_golden(golden, dev):
    """Device-side bounded Brent against the reference's scipy-driven search on the reference-generated fixture.
    The one difference left is the order of the fp32 sum inside the loss's mean (torch's is the build machine's,
    the kernels sum exactly and round once, like the oracle): ranges within 5e-4, evaluation counts within 35 %.
    Against the ORACLE the kernels are exact: test_msefast_equals_oracle."""
    from outlier_suppression_amd.quantization.quantized_module import ObserverDict
    g = golden("msefast")
    for k in range(int(g["n"])):
        cls, bit, sym, ch_axis, reps, nfev, osd = (str(v) for v in g[f"c{k}_info"])
        bit, sym, ch_axis, reps, nfev = int(bit), bool(int(sym)), int(ch_axis), int(reps), int(nfev)
        ob = ObserverDict[cls](bit=bit, symmetric=sym, ch_axis=ch_axis).to(dev)
        x = g[f"c{k}_x"]
        total = 0
        for r in range(reps):
            ret = ob(T(x[r] if reps > 1 else x, dev))
            assert ret is None
            total += int(ob.last_nfev.sum().item())
            np.testing.assert_allclose(N(ob.min_val), g[f"c{k}_min"][r], rtol=5e-4, atol=1e-6)
            np.testing.assert_allclose(N(ob.max_val), g[f"c{k}_max"][r], rtol=5e-4, atol=1e-6)
        assert ob.one_side_dist == osd
        assert ob.min_val.dtype == (torch.float64 if ch_axis == -1 else torch.float32)
        assert abs(total - nfev) <= max(6, 0.35 * nfev), (cls, total, nfev)


def test_msefast_equals_oracle(dev, sum_tier):
    """Iterate for iterate: per-channel rows of BERT-base lengths (768, 3072), short and ragged rows, a row longer than
    the register cache; per-tensor 1-D and nested 2-D searches over a masked activation, two batches each (the
    second one runs on a float64 copy of x in the reference, observer.py:524,549).  Ranges bit-equal, evaluation
    counts equal."""
    from outlier_suppression_amd.quantization.observer import MSEFastObserver, AvgMSEFastObserver
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(8)
    # (rows are summed in the reference's order on both sides, observer_oracle.ROW_SUM_VEC / "mse_rows_order": 8 lanes; 4096
    # and 9000 columns take the long-row form of that kernel, 5 and 7 columns ATen's scalar path)
    for cols, bit, rows in ((768, 4, 48), (3072, 4, 24), (200, 6, 24), (3100, 4, 6), (96, 4, 24), (768, 6, 16), (4096, 4, 6),
                            (9000, 4, 3), (5, 4, 12), (7, 6, 12), (8, 4, 12)):
        w = torch.randn(rows, cols, generator=gen) * 0.05
        ob = MSEFastObserver(bit=bit, symmetric=True, ch_axis=0).to(dev)
        ob(w.to(dev))
        st = OB.ObserverState(bit=bit, symmetric=True, ch_axis=0)
        counter = [0]
        OB.observe_msefast(st, w.numpy(), counter=counter)
        assert bits_equal(N(ob.max_val), st.max_val) and bits_equal(N(ob.min_val), st.min_val), (cols, bit)
        assert int(ob.last_nfev.sum().item()) == counter[0], (cols, bit)
    # one-sided weights (post-ReLU style rows) and asymmetric per-channel (nested search per row)
    w = torch.rand(12, 256, generator=gen) * 0.3
    for sym, data in ((False, w), (False, -w), (False, torch.randn(8, 128, generator=gen) * 0.1)):
        ob = MSEFastObserver(bit=4, symmetric=sym, ch_axis=0).to(dev)
        ob(data.to(dev))
        st = OB.ObserverState(bit=4, symmetric=sym, ch_axis=0)
        OB.observe_msefast(st, data.numpy())
        assert bits_equal(N(ob.max_val), st.max_val) and bits_equal(N(ob.min_val), st.min_val)
    x = torch.randn(8, 32, 96, generator=gen)
    x[..., 3] *= 12
    L = torch.randint(4, 33, (8,), generator=gen)
    for cls, avg in ((AvgMSEFastObserver, True), (MSEFastObserver, False)):
        for sym in (True, False):
            for masked in (True, False):
                ob = cls(bit=6, symmetric=sym).to(dev)
                st = OB.ObserverState(bit=6, symmetric=sym)
                for it in range(3):
                    xi = x * (it + 1)
                    counter = [0]
                    if masked:
                        ob(xi.to(dev), L.to(dev), 1)
                        OB.observe_msefast(st, xi.numpy(), L.numpy(), 1, average=avg, counter=counter)
                    else:
                        ob(xi.to(dev))
                        OB.observe_msefast(st, xi.numpy(), average=avg, counter=counter)
                    if it == 0:      # fp32 arithmetic: exact
                        assert float(N(ob.min_val)) == float(st.min_val) and float(N(ob.max_val)) == float(st.max_val)
                        assert int(ob.last_nfev.sum().item()) == counter[0], (cls.__name__, sym, masked)
                    else:
                        # float64 arithmetic from the second call on (observer.py:524,549).  A float64 sum of float64
                        # squares carries its order in its last bits, and near the minimum of a staircase loss
                        # candidates tie to that precision: mostly 1e-12 apart, now and then a neighbouring step -- and
                        # the nested 2-D search amplifies it.  Measured on these very inputs with the ORACLE alone,
                        # summing the same squares pairwise (NumPy) / in torch's order / left to right / right to
                        # left: up to 7e-4 between orders for the 1-D search, up to 1.3e-2 for the 2-D one.
                        np.testing.assert_allclose(N(ob.min_val), st.min_val, rtol=2e-3 if sym else 3e-2)
                        np.testing.assert_allclose(N(ob.max_val), st.max_val, rtol=2e-3 if sym else 3e-2)
                assert (not avg) or ob.cnt == 3
    # the float64 arithmetic on its own: observers whose statistics are float64 before their first call, fresh per batch,
    # so that the search result itself (not a running statistic) is compared -- and, where a tie broke the other way,
    # the objective: the kernel's range must be as good as the oracle's
    for sym in (True, False):
        for it in range(3):
            xi = x * (1.0 + 0.37 * it)
            ob = MSEFastObserver(bit=6, symmetric=sym).to(dev)
            ob.min_val, ob.max_val = ob.min_val.double(), ob.max_val.double()
            st = OB.ObserverState(bit=6, symmetric=sym)
            st.min_val, st.max_val = np.asarray(np.float64(np.inf)), np.asarray(np.float64(-np.inf))
            ob(xi.to(dev), L.to(dev), 1)
            OB.observe_msefast(st, xi.numpy(), L.numpy(), 1)
            v = OB.remove_padding(xi.numpy(), L.numpy(), 1).astype(np.float64)
            ours = OB.mse_loss(v, float(N(ob.min_val)), float(N(ob.max_val)), st.quant_min, st.quant_max, sym)
            ref = OB.mse_loss(v, float(st.min_val), float(st.max_val), st.quant_min, st.quant_max, sym)
            assert ours <= ref * (1 + 1e-3), (sym, it, ours, ref)
            np.testing.assert_allclose(N(ob.max_val), st.max_val, rtol=2e-3 if sym else 3e-2)


def test_msefast_float64_equals_oracle_with_exact_sums(dev):
    """The float64 branch of per-tensor MSEFast (from an observer's second call on the reference searches on a float64 copy
    of x, observer.py:524,549) pinned at the bit level: with the loss summed order-independently on both sides -- the
    kernels' double-double test mode, osq_set_tuning("mse_sum_order", 64), and math.fsum in the oracle -- ranges, running
    means and evaluation counts of EVERY batch are equal, not just close: 1-D and nested 2-D searches, Avg and plain
    observers, masked and flat inputs.  What test_msefast_equals_oracle tolerates (2e-3 / 3e-2) is therefore the rounding
    noise of a plain float64 sum and nothing else in the port; the same noise separates the reference from both."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import MSEFastObserver, AvgMSEFastObserver
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(18)
    x = torch.randn(8, 32, 96, generator=gen)
    x[..., 3] *= 12
    L = torch.randint(4, 33, (8,), generator=gen)
    ops.set_tuning("mse_sum_order", 64)
    old_mean = OB.MEAN_LIKE_TORCH
    OB.MEAN_LIKE_TORCH = OB.exact_mean
    try:
        for cls, avg in ((AvgMSEFastObserver, True), (MSEFastObserver, False)):
            for sym in (True, False):
                for masked in (True, False):
                    ob = cls(bit=6, symmetric=sym).to(dev)
                    st = OB.ObserverState(bit=6, symmetric=sym)
                    for it in range(3):
                        xi = x * (1.0 + 0.6 * it)
                        counter = [0]
                        if masked:
                            ob(xi.to(dev), L.to(dev), 1)
                            OB.observe_msefast(st, xi.numpy(), L.numpy(), 1, average=avg, counter=counter)
                        else:
                            ob(xi.to(dev))
                            OB.observe_msefast(st, xi.numpy(), average=avg, counter=counter)
                        what = (cls.__name__, sym, masked, it)
                        assert float(N(ob.min_val)) == float(st.min_val) and float(N(ob.max_val)) == float(st.max_val), \
                            (what, float(N(ob.min_val)), float(st.min_val), float(N(ob.max_val)), float(st.max_val))
                        assert int(ob.last_nfev.sum().item()) == counter[0], what
    finally:
        OB.MEAN_LIKE_TORCH = old_mean
        ops.set_tuning("mse_sum_order", 8)      # the default


def test_mse_grid_equals_oracle(dev):
    """MSEObserver / AvgMSEObserver (brute-force grid, observer.py:285-409): the device search and the oracle evaluate the
    same candidates, sum the same fp32 squared errors in float64 and round the mean to fp32 once, so the argmin -- first
    best candidate wins -- is the same grid point: ranges bit-equal.  (Against the reference, whose loss is an fp32
    torch sum in the build machine's order, near-ties may pick the neighbouring grid point: test_other_observers_golden.)
    1-D (symmetric, one-sided) and 2-D (asymmetric) searches, per-tensor masked / unmasked and per-channel rows."""
    from outlier_suppression_amd.quantization.quantized_module import ObserverDict
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(6, 24, 64, generator=gen)
    x[..., 3] *= 7
    L = torch.randint(3, 25, (6,), generator=gen)
    for cls in ("MSEObserver", "AvgMSEObserver"):
        for sym, data in ((True, x), (False, x), (False, x.abs()), (False, -x.abs())):
            for masked in (False, True):
                ob = ObserverDict[cls](bit=6, symmetric=sym).to(dev)
                st = OB.ObserverState(bit=6, symmetric=sym)
                for it in range(2):
                    xi = data * (1.0 + 0.3 * it)
                    if masked:
                        ob(xi.to(dev), L.to(dev), 1)
                        OB.observe_mse(st, xi.numpy(), L.numpy(), 1, average=cls.startswith("Avg"))
                    else:
                        ob(xi.to(dev))
                        OB.observe_mse(st, xi.numpy(), average=cls.startswith("Avg"))
                    assert bits_equal(N(ob.min_val), np.asarray(st.min_val, dtype=np.float32)) and \
                        bits_equal(N(ob.max_val), np.asarray(st.max_val, dtype=np.float32)), (cls, sym, masked, it, N(ob.min_val), st.min_val, N(ob.max_val), st.max_val)
    w = torch.randn(12, 96, generator=gen) * 0.05
    for sym, data in ((True, w), (False, w), (False, w.abs())):
        ob = ObserverDict["MSEObserver"](bit=4, symmetric=sym, ch_axis=0).to(dev)
        st = OB.ObserverState(bit=4, symmetric=sym, ch_axis=0)
        ob(data.to(dev))
        OB.observe_mse(st, data.numpy())
        assert bits_equal(N(ob.min_val), st.min_val) and bits_equal(N(ob.max_val), st.max_val), (sym,)


def test_msefast_resident_search_equals_launch_per_evaluation(dev):
    """The per-tensor search has two forms: one persistent launch with the valid part of the tensor in the grid's
    registers (default when it fits), and one launch per loss evaluation.  Both run the same state machine on the same
    squared errors, summed in a different order: in float32 arithmetic (first call of an observer) the mean is rounded
    to fp32 once and the searches must agree exactly -- range and evaluation count; in float64 arithmetic (later calls)
    the sum's order is in its last bits (see test_msefast_equals_oracle): same evaluation count or a range within the
    oracle's own order-sensitivity.  Layouts: flat with a tail, dense masked, head-split view, strided key view (scalar
    gathers), 3-D probabilities with the zip truncation, a tensor that does NOT fit (falls back by itself)."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver, MSEFastObserver
    gen = torch.Generator().manual_seed(31)
    B, Tn = 8, 48
    L = torch.randint(1, Tn + 1, (B,), generator=gen)
    L[0] = Tn
    L[3] = 0
    base = torch.randn(B, Tn, 96, generator=gen)
    base[..., 3] *= 9
    cases = [
        ("flat+tail", torch.randn(4099, generator=gen) * 2, None, -1),
        ("dense masked", base, L, 1),
        ("head split", base.view(B, Tn, 4, 24).permute(0, 2, 1, 3), L, 2),
        ("key view", base.view(B, Tn, 4, 24).permute(0, 2, 3, 1), L, 3),
        ("probs 3-D", torch.rand(4 * B, Tn, Tn, generator=gen), L, 1),
        ("one-sided", base.abs(), L, 1),
        ("too large for the registers", torch.randn(40, 512, 1024, generator=gen), torch.randint(1, 513, (40,), generator=gen), 1),
    ]
    for name, x, mask, seq_pos in cases:
        for cls in (AvgMSEFastObserver, MSEFastObserver):
            for sym in (False, True):
                got = {}
                for resident in (1, 0):
                    ops.set_tuning("mse_resident", resident)
                    try:
                        ob = cls(bit=6, symmetric=sym).to(dev)
                        rec = []
                        for it in range(2 if x.numel() < (1 << 22) else 1):
                            xi = (x * (1.0 + 0.5 * it)).to(dev)
                            if mask is None:
                                ob(xi)
                            else:
                                ob(xi, mask.to(dev), seq_pos)
                            rec.append((float(N(ob.min_val)), float(N(ob.max_val)), int(ob.last_nfev.sum().item())))
                        got[resident] = rec
                    finally:
                        ops.set_tuning("mse_resident", 1)
                (a0, b0) = got[1][0], got[0][0]
                assert a0 == b0, (name, cls.__name__, sym, a0, b0)                    # float32 arithmetic: exact
                for a, b in zip(got[1][1:], got[0][1:]):                             # float64 arithmetic
                    assert a[2] == b[2] or abs(a[2] - b[2]) <= 0.35 * b[2], (name, cls.__name__, sym, a, b)
                    np.testing.assert_allclose(a[:2], b[:2], rtol=2e-3 if sym else 3e-2, atol=1e-9, err_msg=f"{name} {cls.__name__} {sym}")


def test_resident_search_time_out_is_loud_and_recoverable(dev, order_free):
    """A resident MSEFast search whose workgroups do not meet (the test knob makes every collection of partial sums give up
    at once) poisons its range with NaN, the commit PROPAGATES the NaN into min_val / max_val / scale (round 2's commit
    dropped it: `bmin < mn ? bmin : mn`), and the host raises at the call's own synchronisation point; the next search on
    the same workspace is correct."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import MSEFastObserver
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(16, 64, 96, generator=gen) * 3).to(dev)
    L = torch.randint(1, 65, (16,), generator=gen).to(dev)
    ops.check_persistent()
    good = MSEFastObserver(bit=6, symmetric=False).to(dev)
    good(x, L, 1)
    want = (good.min_val.item(), good.max_val.item())
    ops.set_tuning("mse_spin_limit", 1)
    try:
        ob = MSEFastObserver(bit=6, symmetric=False).to(dev)
        with pytest.raises(ops.PersistentLaunchTimeout):
            ob(x, L, 1)
        assert torch.isnan(ob.min_val).all() and torch.isnan(ob.max_val).all()
        # several searches in one launch (an observer pass): the flush raises
        from types import SimpleNamespace as NS
        from outlier_suppression_amd.quantization import Quantizer
        qs = [Quantizer(None, NS(quantizer="FixedFakeQuantize", observer="AvgMSEFastObserver", bit=6, symmetric=False, ch_axis=-1)).to(dev)
              for _ in range(3)]
        for q in qs:
            q.enable_observer()
        with pytest.raises(ops.PersistentLaunchTimeout):
            with deferred_observation() as sites:
                for q in qs:
                    q(x, L, 1)
                sites.flush()
        assert all(torch.isnan(q.scale).all() for q in qs)
    finally:
        ops.set_tuning("mse_spin_limit", 0)
    ops.check_persistent()
    again = MSEFastObserver(bit=6, symmetric=False).to(dev)
    again(x, L, 1)
    assert (again.min_val.item(), again.max_val.item()) == want


def test_msefast_searches_of_a_forward_share_a_launch(dev):
    """Inside deferred_observation() the per-tensor MSEFast searches of a forward are recorded and run together, up to 16
    per persistent launch (every round evaluates every unfinished search).  Each search does the arithmetic it does
    alone -- same threads, same elements, same order of every sum: ranges, running statistics, scales and evaluation
    counts are bit-equal to the immediate path, in float32 arithmetic (first batch) and in float64 (later batches).
    23 sites of mixed layouts and sizes (several launches; one site too large to be resident), 3 batches."""
    from outlier_suppression_amd.quantization import Quantizer
    from outlier_suppression_amd.quantization.deferred import deferred_observation
    gen = torch.Generator().manual_seed(77)
    B, Tn = 8, 40
    L = torch.randint(1, Tn + 1, (B,), generator=gen)
    L[0] = Tn
    base = torch.randn(B, Tn, 96, generator=gen)
    base[..., 3] *= 9
    def site_inputs(scale):
        x = base * scale
        return [(x, L, 1), (x.view(B, Tn, 4, 24).permute(0, 2, 1, 3), L, 2), (x.view(B, Tn, 4, 24).permute(0, 2, 3, 1), L, 3),
                (torch.rand(2 * B, Tn, Tn, generator=gen) * scale, L, 1), (x.abs(), L, 1), (torch.randn(4099, generator=gen) * scale, None, -1)]
    cfgs = [NS(quantizer="FixedFakeQuantize", observer=o, bit=6, symmetric=s, ch_axis=-1)
            for o in ("AvgMSEFastObserver", "MSEFastObserver") for s in (False, True)]
    results = {}
    for deferred in (False, True):
        gen.manual_seed(123)           # both passes draw the same tensors
        qs = []
        big = torch.randn(40, 512, 1024, generator=torch.Generator().manual_seed(5))       # 84 MB: not resident
        big_L = torch.randint(1, 513, (40,), generator=torch.Generator().manual_seed(6))
        for it in range(3):
            inputs = site_inputs(1.0 + 0.4 * it)
            calls = [(inputs[k % len(inputs)], cfgs[k % len(cfgs)]) for k in range(22)]
            if it == 0:
                qs = [Quantizer(None, c).to(dev) for _, c in calls] + [Quantizer(None, cfgs[0]).to(dev)]
                for q in qs:
                    q.enable_observer()
                    q.disable_fake_quant()
            def forward():
                for q, ((x, mask, sp), _) in zip(qs, calls):
                    q(x.to(dev), None if mask is None else mask.to(dev), sp)
                if it == 0:
                    qs[-1](big.to(dev), big_L.to(dev), 1)
            if deferred:
                with deferred_observation() as sites:
                    forward()
                    sites.flush()
                assert sites.flushed_sites == (23 if it == 0 else 22) and sites.launches < 10
            else:
                forward()
        torch.cuda.synchronize()
        results[deferred] = [(q.observer.min_val.clone(), q.observer.max_val.clone(), q.scale.detach().clone(), q.zero_point.detach().clone(),
                              int(q.observer.last_nfev.sum().item())) for q in qs]
    for k, (a, b) in enumerate(zip(results[False], results[True])):
        assert all(torch.equal(u, v) for u, v in zip(a[:4], b[:4])) and a[4] == b[4], (k, a, b)


def test_msefast_tensor_equals_reference_in_its_summation_order(golden, dev):
    """osq_set_tuning("mse_sum_order", 8): the per-tensor searches add their squared errors in the order torch's CPU kernel
    added them on the fixture machine -- fp32 in 8 lanes on the first call, float64 in 4 lanes from an observer's second
    call on (observer.py:524,549) -- and every statistic after every call of tests/golden/msefast.npz (the reference's
    own run: 1-D symmetric, one-sided, nested 2-D, per-channel rows, two three-batch Avg sequences) is reproduced BIT FOR
    BIT, with the reference's number of loss evaluations."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import observer as OBS
    g = golden("msefast")
    ops.set_tuning("mse_sum_order", 8)
    try:
        for k in range(int(g["n"])):
            cls, bit, sym, ch_axis, reps, nfev, osd = (str(v) for v in g[f"c{k}_info"])
            bit, sym, ch_axis, reps, nfev = int(bit), bool(int(sym)), int(ch_axis), int(reps), int(nfev)
            ob = getattr(OBS, cls)(bit=bit, symmetric=sym, ch_axis=ch_axis).to(dev)
            x = g[f"c{k}_x"]
            evals = 0
            for r in range(reps):
                ob(torch.from_numpy(x[r] if reps > 1 else x).to(dev))
                evals += int(ob.last_nfev.sum().item())
                got_min, got_max = N(ob.min_val).reshape(-1), N(ob.max_val).reshape(-1)
                assert bits_equal(got_min.astype(np.float64), g[f"c{k}_min"][r].reshape(-1).astype(np.float64)) and \
                    bits_equal(got_max.astype(np.float64), g[f"c{k}_max"][r].reshape(-1).astype(np.float64)), \
                    (k, cls, r, got_min, g[f"c{k}_min"][r], got_max, g[f"c{k}_max"][r])
            assert evals == nfev, (k, cls, evals, nfev)
    finally:
        ops.set_tuning("mse_sum_order", 8)      # the default


def test_msefast_masked_tensor_equals_reference_in_its_summation_order(golden, dev):
    """The same test mode on MASKED activations (tests/golden/msefast_masked.npz: observation_mask + seq_pos, one case a
    [B,h,T,d] tensor with the tokens on axis 2; three batches each, the later ones searched in float64): the kernel lays the
    squared errors out in remove_padding's order (observer.py:72-84) and every statistic after every call, and the number
    of loss evaluations, equal the reference's run bit for bit."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import observer as OBS
    g = golden("msefast_masked")
    ops.set_tuning("mse_sum_order", 8)
    try:
        for k in range(int(g["n"])):
            cls, bit, sym, seq_pos, nfev, osd = (str(v) for v in g[f"c{k}_info"])
            ob = getattr(OBS, cls)(bit=int(bit), symmetric=bool(int(sym)), ch_axis=-1).to(dev)
            evals = 0
            for r in range(3):
                ob(torch.from_numpy(g[f"c{k}_x"][r]).to(dev), torch.from_numpy(g[f"c{k}_len"][r]).to(dev), int(seq_pos))
                evals += int(ob.last_nfev.sum().item())
                assert bits_equal(N(ob.min_val).reshape(-1).astype(np.float64), g[f"c{k}_min"][r].reshape(-1)) and \
                    bits_equal(N(ob.max_val).reshape(-1).astype(np.float64), g[f"c{k}_max"][r].reshape(-1)), \
                    (k, cls, r, N(ob.min_val), g[f"c{k}_min"][r], N(ob.max_val), g[f"c{k}_max"][r])
            assert evals == int(nfev), (k, cls, evals, nfev)
    finally:
        ops.set_tuning("mse_sum_order", 8)      # the default


@pytest.mark.parametrize("name", ["w768", "w3072", "w768_6bit"])
def test_msefast_rows_against_reference(golden, name, dev):
    """The ORDER-FREE variant (osq_set_tuning("mse_rows_order", 0): the loss summed exactly and rounded once) on every row of
    the reference-generated fixture (2048 rows of 768 and of 3072 columns at 4 bit, 1024 rows at 6 bit;
    tests/golden/make_golden.py::gen_msefast_rows): how far its ranges are from the reference's, and what that does to the
    integers.  Measured: median relative range error 2e-5, 99th percentile 2e-4, x_quant entries that differ
    1.5e-5 .. 4e-5 of all entries -- which is why the DEFAULT sums rows in the reference's order
    (test_msefast_rows_equal_reference_in_its_summation_order: bit-equal)."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import MSEFastObserver
    from _msefast_rows import MSEFAST_ROW_BOUNDS, msefast_row_weights
    g = golden("msefast_rows")
    seed, rows, cols, bit, ref_nfev = (int(v) for v in g[name + "_info"])
    w = torch.from_numpy(msefast_row_weights(seed, rows, cols)).to(dev)
    ob = MSEFastObserver(bit=bit, symmetric=True, ch_axis=0).to(dev)
    ops.set_tuning("mse_rows_order", 0)
    try:
        ob(w)
        torch.cuda.synchronize()
    finally:
        ops.set_tuning("mse_rows_order", 8)
    ref_min, ref_max = T(g[name + "_min"], dev), T(g[name + "_max"], dev)
    rel = ((ob.max_val - ref_max).abs() / ref_max).cpu().numpy()
    b = MSEFAST_ROW_BOUNDS
    assert np.median(rel) <= b["median_rel"] and np.quantile(rel, 0.99) <= b["p99_rel"] and rel.max() <= b["max_rel"], \
        (np.median(rel), np.quantile(rel, 0.99), rel.max())
    qmin, qmax = ob.quant_min, ob.quant_max
    s_a, z_a = ob.calculate_qparams(ob.min_val, ob.max_val)
    s_b, z_b = ob.calculate_qparams(ref_min, ref_max)
    _, xa = ops.fake_quant_per_channel(w, s_a, z_a, 0, qmin, qmax, return_quantized=True)
    _, xb = ops.fake_quant_per_channel(w, s_b, z_b, 0, qmin, qmax, return_quantized=True)
    mismatch = (xa != xb).float().mean().item()
    assert mismatch <= b["xquant_mismatch"], mismatch
    assert abs(int(ob.last_nfev.sum().item()) - ref_nfev) <= 0.02 * ref_nfev


@pytest.mark.parametrize("name", ["w768", "w3072", "w768_6bit"])
def test_msefast_rows_equal_reference_in_its_summation_order(golden, name, dev):
    """The DEFAULT per-row kernel adds its squared errors in the order torch's CPU kernel adds them (8 SIMD lanes, 4
    interleaved cascades per lane; restated in oracle/aten_sum.py and pinned against torch.sum; a row is shorter than
    ATen's parallel grain, so that order is the reference's whatever the host's thread count).  Every one of the fixture's
    5120 reference-generated rows comes out BIT-EQUAL -- min_val, max_val, hence every scale and every x_quant entry --
    and the evaluation count equals the reference's exactly."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import MSEFastObserver
    from _msefast_rows import msefast_row_weights
    g = golden("msefast_rows")
    seed, rows, cols, bit, ref_nfev = (int(v) for v in g[name + "_info"])
    w = torch.from_numpy(msefast_row_weights(seed, rows, cols)).to(dev)
    ob = MSEFastObserver(bit=bit, symmetric=True, ch_axis=0).to(dev)
    ob(w)
    torch.cuda.synchronize()
    assert bits_equal(N(ob.max_val), g[name + "_max"]) and bits_equal(N(ob.min_val), g[name + "_min"])
    assert int(ob.last_nfev.sum().item()) == ref_nfev
    s_a, z_a = ob.calculate_qparams(ob.min_val, ob.max_val)
    s_b, z_b = ob.calculate_qparams(T(g[name + "_min"], dev), T(g[name + "_max"], dev))
    _, xa = ops.fake_quant_per_channel(w, s_a, z_a, 0, ob.quant_min, ob.quant_max, return_quantized=True)
    _, xb = ops.fake_quant_per_channel(w, s_b, z_b, 0, ob.quant_min, ob.quant_max, return_quantized=True)
    assert torch.equal(xa, xb)


def test_msefast_through_quantizer(dev):
    from outlier_suppression_amd.quantization import Quantizer
    from oracle import observer_oracle as OB
    cfg = NS(quantizer="FixedFakeQuantize", observer="MSEFastObserver", bit=4, symmetric=True, ch_axis=0)
    torch.manual_seed(5)
    lin = torch.nn.Linear(96, 10)
    ql = Quantizer(lin, cfg).to(dev)
    ql.weight_fake_quant.enable_observer()
    ql.weight_fake_quant.enable_fake_quant()
    x = torch.randn(3, 96, device=dev)
    y = ql(x)
    assert y.shape == (3, 10)
    fq = ql.weight_fake_quant
    st = OB.ObserverState(bit=4, symmetric=True, ch_axis=0)
    OB.observe_msefast(st, lin.weight.detach().numpy())
    s_o, _ = st.qparams()
    assert bits_equal(N(fq.observer.max_val), st.max_val) and bits_equal(N(fq.scale), s_o)
    assert fq.scale.shape == (10,) and fq.zero_point.dtype == torch.int32


def test_wide_finaliser_vs_oracle(eq32, dev):
    """>= 8192 token slots take the three-launch multi-workgroup finaliser: prune and plain paths, a
    narrow distribution (every key in ONE coarse bin -> the list holds all tokens), heavy duplicates,
    repeated calls (global scratch must come back zeroed)."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import AvgPruneMinMaxObserver, AvgMinMaxObserver
    from oracle import observer_oracle as OB
    ops.set_wide_min_slots(8192)    # default 32769: exercise the wide path at a size the oracle handles quickly
    gen = torch.Generator().manual_seed(77)
    B, Tn, H = 96, 128, 64          # 12288 slots
    cases = {
        "spread": lambda: torch.randn(B, Tn, H, generator=gen) * torch.rand(B, Tn, 1, generator=gen) * 9,
        "narrow": lambda: 4.0 + torch.rand(B, Tn, H, generator=gen) * 0.4,            # all token maxima in [4.0, 4.5)
        "dups": lambda: torch.randint(-3, 4, (B, Tn, H), generator=gen).float(),      # a handful of distinct extrema
    }
    for name, make in cases.items():
        for cls, fn, p in ((AvgPruneMinMaxObserver, OB.observe_avg_prune_minmax, 0.9),
                           (AvgPruneMinMaxObserver, OB.observe_avg_prune_minmax, 1.0),
                           (AvgMinMaxObserver, OB.observe_avg_minmax, None)):
            ob = cls(bit=6).to(dev)
            ob.set_name("x_post_act_fake_quantize.observer")
            st = OB.ObserverState(bit=6, name=ob.name)
            if p is not None:
                ob.set_percentile(p)
                st.percentile = p
            for it in range(2):
                x = make()
                L = torch.randint(1, Tn + 1, (B,), generator=gen)
                ob(x.to(dev), L.to(dev), 1)
                fn(st, x.numpy(), L.numpy(), 1)
                assert eq32(N(ob.min_val), st.min_val) and eq32(N(ob.max_val), st.max_val), (name, cls.__name__, p, it)
    # NaN poisons, and the scratch is clean afterwards
    x = torch.randn(B, Tn, H, generator=gen)
    x[3, 0, 5] = float("nan")
    L = torch.full((B,), Tn)
    ob = AvgPruneMinMaxObserver(bit=6).to(dev)
    ob.set_name("x")
    ob.set_percentile(0.9)
    ob(x.to(dev), L.to(dev), 1)
    assert torch.isnan(ob.min_val).item()
    ob2 = AvgPruneMinMaxObserver(bit=6).to(dev)
    ob2.set_name("x")
    ob2.set_percentile(0.9)
    x[3, 0, 5] = 0.0
    ob2(x.to(dev), L.to(dev), 1)
    st = OB.ObserverState(bit=6, name="x")
    st.percentile = 0.9
    OB.observe_avg_prune_minmax(st, x.numpy(), L.numpy(), 1)
    assert eq32(N(ob2.min_val), st.min_val) and eq32(N(ob2.max_val), st.max_val)
    ops.set_wide_min_slots(32769)


def test_token_minmax_placements(dev):
    """Per-token extrema on the XCD-rotated (chunk, sample) grid: identical to stock amin/amax on every valid
    token, padded slots untouched; lengths with zeros, full rows, T not a multiple of 16, head-split views.
    (A compacted placement -- workgroup i takes the i-th valid chunk via a block-wide prefix sum -- was tried
    and measured no faster: 13.2 vs 13.0 us on the 54 %-valid [256,128,768] tensor.)"""
    from outlier_suppression_amd import ops
    gen = torch.Generator().manual_seed(5)
    cases = [((200, 100, 64), 1), ((1000, 33, 32), 1), ((1100, 16, 32), 1), ((160, 4, 128, 16), 2), ((8, 40, 64), 1)]
    for shape, sp in cases:
        x = torch.randn(*shape, generator=gen).to(dev)
        Tn = shape[sp]
        L = torch.randint(0, Tn + 1, (shape[0],), generator=gen)
        L[0], L[-1] = Tn, 0
        L = L.to(dev)
        n = shape[0] * Tn
        sentinel = 123456.0
        out = (torch.full((n,), sentinel, device=dev), torch.full((n,), sentinel, device=dev))
        tmin, tmax, B, T_, _ = ops.token_minmax(x, sp, L, out=out)
        feat = [d for d in range(x.dim()) if d not in (0, sp)]
        ref_min = x.amin(dim=feat).reshape(B, T_)
        ref_max = x.amax(dim=feat).reshape(B, T_)
        valid = torch.arange(T_, device=dev)[None, :] < L[:, None]
        assert torch.equal(tmin.view(B, T_)[valid], ref_min[valid]), shape
        assert torch.equal(tmax.view(B, T_)[valid], ref_max[valid]), shape
        assert (tmin.view(B, T_)[~valid] == sentinel).all() and (tmax.view(B, T_)[~valid] == sentinel).all(), shape


def test_finaliser_paths_agree_with_oracle(eq32, dev):
    """The token range finaliser has three implementations -- two workgroups (one per side, the default
    up to 32768 slots), the single-workgroup kernel (layouts the fast one rejects) and the three-launch
    wide one -- which must all reproduce the oracle bit for bit: geometries with T % 4 != 0 (16-byte
    groups straddling two samples), empty samples, one valid token, duplicates, narrow ranges,
    percentiles 0 / 1 / in between, repeated calls (the rendezvous word must come back to zero)."""
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization.observer import AvgPruneMinMaxObserver, AvgMinMaxObserver
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(123)
    geoms = [(8, 6, 16), (8, 5, 8), (4, 4, 8), (32, 128, 32), (3, 40, 64), (64, 128, 16), (16, 36, 8), (1, 4, 4)]
    makers = {
        "spread": lambda B, Tn, H: torch.randn(B, Tn, H, generator=gen) * torch.rand(B, Tn, 1, generator=gen) * 9,
        "narrow": lambda B, Tn, H: 4.0 + torch.rand(B, Tn, H, generator=gen) * 0.4,
        "dups": lambda B, Tn, H: torch.randint(-3, 4, (B, Tn, H), generator=gen).float(),
        "negative": lambda B, Tn, H: -1.0 - torch.rand(B, Tn, H, generator=gen) * 5,
    }
    def run(path):
        ops.set_tuning("final_fast", 0 if path == "single" else 1)
        ops.set_tuning("select_shortcut", 0 if path == "select_full_pass" else 1)
        ops.set_wide_min_slots(1024 if path == "wide" else 32769)
        try:
            for (B, Tn, H) in geoms:
                for name, make in makers.items():
                    for cls, fn, p in ((AvgPruneMinMaxObserver, OB.observe_avg_prune_minmax, 0.9),
                                       (AvgPruneMinMaxObserver, OB.observe_avg_prune_minmax, 1.0),
                                       (AvgPruneMinMaxObserver, OB.observe_avg_prune_minmax, 0.0),
                                       (AvgPruneMinMaxObserver, OB.observe_avg_prune_minmax, 0.5),
                                       (AvgMinMaxObserver, OB.observe_avg_minmax, None)):
                        ob = cls(bit=6).to(dev)
                        ob.set_name("x_post_act_fake_quantize.observer")
                        st = OB.ObserverState(bit=6, name=ob.name)
                        if p is not None:
                            ob.set_percentile(p)
                            st.percentile = p
                        for it in range(3):
                            x = make(B, Tn, H)
                            L = torch.randint(0 if it == 1 else 1, Tn + 1, (B,), generator=gen)
                            if it == 2:
                                L[:] = 0
                                L[B // 2] = 1                  # a single valid token in the whole batch
                            else:
                                L[0] = Tn
                            ob(x.to(dev), L.to(dev), 1)
                            fn(st, x.numpy(), L.numpy(), 1)
                            assert eq32(N(ob.min_val), st.min_val) and eq32(N(ob.max_val), st.max_val), \
                                (path, (B, Tn, H), name, cls.__name__, p, it)
            # all-padding batch: nothing observed, state untouched; then a NaN among the valid tokens poisons
            ob = AvgPruneMinMaxObserver(bit=6).to(dev)
            ob.set_name("x")
            ob.set_percentile(0.9)
            x = torch.randn(8, 16, 32, generator=gen)
            ob(x.to(dev), torch.zeros(8, dtype=torch.int64, device=dev), 1)
            assert torch.isinf(ob.max_val).item() and torch.isinf(ob.min_val).item()
            x[2, 1, 3] = float("nan")
            ob(x.to(dev), torch.full((8,), 16, device=dev), 1)
            assert torch.isnan(ob.min_val).item() and torch.isnan(ob.max_val).item()
        finally:
            ops.set_tuning("final_fast", 1)
            ops.set_tuning("select_shortcut", 1)
            ops.set_wide_min_slots(32769)
    for path in ("select", "select_full_pass", "single", "wide"):
        run(path)


def test_select_shortcut_equals_full_pass(dev):
    """token_select_kernel answers max(v[v <= thr]) from the candidate list when an element with the lower
    order-statistic key is non-negative, and by a pass over the registers otherwise: both must give the same bits
    on mixed signs, heavy duplicates, tiny N, thresholds that land exactly on the upper key, all percentiles."""
    from outlier_suppression_amd import ops
    gen = torch.Generator().manual_seed(2024)
    cur = [torch.empty(2, device=dev), torch.empty(2, device=dev)]
    n_cases = 0
    for trial in range(120):
        B = int(torch.randint(1, 40, (1,), generator=gen))
        T_ = int(torch.randint(1, 20, (1,), generator=gen)) * 4
        kind = trial % 6
        shape = (B * T_,)
        if kind == 0:
            tmax = torch.randn(shape, generator=gen)
        elif kind == 1:
            tmax = torch.randint(-3, 4, shape, generator=gen).float()
        elif kind == 2:
            tmax = -torch.rand(shape, generator=gen) - 0.5
        elif kind == 3:
            tmax = torch.randn(shape, generator=gen).abs() * torch.where(torch.rand(shape, generator=gen) < 0.1, -1.0, 1.0)
        elif kind == 4:
            tmax = torch.randint(0, 2, shape, generator=gen).float() * 2 - 1        # only +1 / -1
        else:
            tmax = torch.randn(shape, generator=gen) * 1e-3 + 5.0
        tmin = tmax - torch.rand(shape, generator=gen) * (1.0 if kind != 1 else 0.0) - (torch.randint(0, 3, shape, generator=gen).float() if kind == 1 else 0.0)
        L = torch.randint(0, T_ + 1, (B,), generator=gen)
        L[int(torch.randint(0, B, (1,), generator=gen))] = T_
        tmin_d, tmax_d, L_d = tmin.to(dev), tmax.to(dev), L.to(dev)
        for p in (0.0, 0.3, 0.5, 0.77, 0.9, 0.95, 0.99, 1.0):
            for k, flag in enumerate((1, 0)):
                ops.set_tuning("select_shortcut", flag)
                ops.token_range_finalize(tmin_d, tmax_d, B, T_, L_d, True, p, ops.UPDATE_NONE, 0, None, None, 0, 63, False,
                                         None, cur[k])
            ops.set_tuning("select_shortcut", 1)
            a, b = cur[0].cpu(), cur[1].cpu()
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (trial, kind, B, T_, p, a, b)
            n_cases += 1
    assert n_cases == 960


def test_select_large_problems_match_oracle(dev):
    """token_select_kernel at 4096 .. 32768 token slots (2 .. 8 sixteen-byte groups per thread) against the oracle,
    bit for bit: i.i.d. values, outlier tokens, heavy duplicates, one constant, values sorted along the slots, all
    negative, with and without masks, T % 4 != 0, percentiles from 0 to 1; the shortcut and the register pass agree."""
    from outlier_suppression_amd import ops
    from oracle import observer_oracle as OB
    gen = torch.Generator().manual_seed(77)
    cur = [torch.empty(2, device=dev), torch.empty(2, device=dev)]
    n_cases = 0
    for (B, T_) in ((256, 128), (64, 101), (36, 130), (128, 64), (16, 384), (32, 128)):
        n = B * T_
        for kind in ("iid", "outliers", "dups", "const", "sorted", "negative"):
            if kind == "iid":
                tmax = torch.randn(n, generator=gen).abs() + 2.0
            elif kind == "outliers":
                tmax = torch.randn(n, generator=gen).abs() + 2.0
                tmax[torch.rand(n, generator=gen) < 0.03] *= 25.0
            elif kind == "dups":
                tmax = torch.randint(0, 7, (n,), generator=gen).float()
            elif kind == "const":
                tmax = torch.full((n,), 3.5)
            elif kind == "sorted":
                tmax = torch.sort(torch.randn(n, generator=gen).abs() + 1.0).values
            else:
                tmax = -torch.rand(n, generator=gen) * 4.0 - 0.25
            tmin = tmax - torch.rand(n, generator=gen) * 3.0 - (torch.randint(0, 5, (n,), generator=gen).float() if kind == "dups" else 0.0)
            for masked in (False, True):
                if masked:
                    L = torch.randint(T_ // 2, T_ + 1, (B,), generator=gen)
                    L[0] = T_
                else:
                    L = torch.full((B,), T_)
                valid = (torch.arange(T_)[None, :] < L[:, None]).reshape(-1)
                tmin_d, tmax_d, L_d = tmin.to(dev), tmax.to(dev), L.to(dev)
                for p in (0.0, 0.3, 0.5, 0.77, 0.9, 0.95, 0.99, 0.999, 1.0):
                    for k, flag in enumerate((1, 0)):
                        ops.set_tuning("select_shortcut", flag)
                        try:
                            ops.token_range_finalize(tmin_d, tmax_d, B, T_, L_d if masked else None, True, p, ops.UPDATE_NONE, 0,
                                                     None, None, 0, 63, False, None, cur[k])
                        finally:
                            ops.set_tuning("select_shortcut", 1)
                    a, b = cur[0].cpu(), cur[1].cpu()
                    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (B, T_, kind, masked, p, a, b)
                    lo, up = OB.prune_thresholds(tmin[valid].numpy(), tmax[valid].numpy(), p)
                    want = np.array([up if lo > up else lo, up], dtype=np.float32)
                    assert bits_equal(a.numpy().view(np.int32), want.view(np.int32)), (B, T_, kind, masked, p, a, want)
                    n_cases += 1
    assert n_cases == 6 * 6 * 2 * 9


# ----------------------------------------------------------------------------------- remaining observers (N3)

def test_other_observers_golden(golden, eq32, dev):
    """LSQPlusObserver (1e-5: float sums), AvgQuantileObserver (bit-exact incl. torch.histc binning),
    MSEObserver / AvgMSEObserver (grid argmin over fp32 losses: within one grid step of the reference)."""
    from outlier_suppression_amd.quantization.quantized_module import ObserverDict
    g = golden("other_observers")
    ob = ObserverDict["LSQPlusObserver"](bit=8, symmetric=True, ch_axis=-1).to(dev)
    ob(T(g["lsqp_x"], dev))
    np.testing.assert_allclose(N(ob.min_val), g["lsqp_min"], rtol=1e-5)
    np.testing.assert_allclose(N(ob.max_val), g["lsqp_max"], rtol=1e-5)
    ob = ObserverDict["LSQPlusObserver"](bit=4, symmetric=True, ch_axis=0).to(dev)
    ob(T(g["lsqp_w"], dev))
    np.testing.assert_allclose(N(ob.min_val), g["lsqp_wmin"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(N(ob.max_val), g["lsqp_wmax"], rtol=1e-5, atol=1e-7)
    s, _ = ob.calculate_qparams(ob.min_val, ob.max_val)
    np.testing.assert_allclose(N(s), g["lsqp_wscale"], rtol=1e-5)
    for k in range(int(g["aq_n"])):
        threshold, masked = float(g[f"aq{k}_meta"][0]), bool(g[f"aq{k}_meta"][1])
        ob = ObserverDict["AvgQuantileObserver"](bit=6, threshold=threshold).to(dev)
        for it in range(3):
            ob(T(g[f"aq{k}_x"][it], dev), T(g[f"aq{k}_len"][it], dev) if masked else None, 1 if masked else -1)
            assert eq32(N(ob.min_val), g[f"aq{k}_min"][it]) and eq32(N(ob.max_val), g[f"aq{k}_max"][it]), (k, it)
        assert ob.cnt == 3
    for k in range(int(g["mse_n"])):
        cls, bit, sym, ch_axis, reps, osd = (str(v) for v in g[f"mse{k}_info"])
        ob = ObserverDict[cls](bit=int(bit), symmetric=bool(int(sym)), ch_axis=int(ch_axis)).to(dev)
        x = g[f"mse{k}_x"]
        for r in range(int(reps)):
            assert ob(T(x[r] if int(reps) > 1 else x, dev)) is None
            # bit-equal on every fixture (the reference's fp32 loss sums pick the same grid point here; a near-tie between two
            # candidates could legitimately land one grid step away -- the exact statement is test_mse_grid_equals_oracle)
            assert eq32(N(ob.min_val), g[f"mse{k}_min"][r]) and eq32(N(ob.max_val), g[f"mse{k}_max"][r]), (k, r)
        assert ob.one_side_dist == osd


def test_histogram_matches_torch_histc(dev):
    """The device binning against torch.histc on fresh data (counts must be identical)."""
    from outlier_suppression_amd import ops
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(8, 64, 96, generator=gen)
    x[..., 7] *= 30
    cur = ops.batch_minmax(x.to(dev))
    hist = torch.zeros(2048, dtype=torch.int32, device=dev)
    from outlier_suppression_amd import _hip
    lib = _hip.load()
    # histogram only: run the full call and read the table back BEFORE it is cleared is not possible, so
    # compare the outcome (clip) for many thresholds instead -- each threshold probes a different prefix
    mx = float(max(-x.min(), x.max()))
    ref_hist = torch.histc(x.abs(), bins=2048, min=0.0, max=mx)
    cum = torch.cumsum(ref_hist, 0)
    for thr in (0.5, 0.9, 0.99, 0.999, 0.9999, 0.99999):
        mn = torch.tensor(float("inf"), device=dev)
        mxv = torch.tensor(float("-inf"), device=dev)
        ops.observe_quantile(x.to(dev), None, -1, cur, thr, hist, ops.UPDATE_AVERAGE, 0, mn, mxv, 0, 63, False)
        target = np.float32(thr * x.numel())
        i = int((cum >= float(target)).nonzero()[0])
        clip = np.float32(np.float32(i + 0.5) * np.float32(np.float32(mx) / np.float32(2048)))
        assert mxv.item() == min(np.float32(x.max().item()), clip), thr
        assert mn.item() == max(np.float32(x.min().item()), -clip), thr
        assert int(hist.sum().item()) == 0


def test_batched_rethreshold_with_per_quantizer_lengths(dev):
    """osq_token_range_finalize_batched: every (quantizer, batch) pair in one launch, each quantizer with its own mask
    (BART: cross-attention keys are masked with the decoder's lengths, quant_bart.py:167,172,472) -- equal to the
    single-problem finaliser called pair by pair; shape errors are refused, not read out of bounds."""
    from outlier_suppression_amd import ops
    gen = torch.Generator().manual_seed(3)
    n_q, n_b, B, T = 3, 4, 8, 16
    tmin = (-torch.rand(n_q, n_b, B * T, generator=gen) * 5).to(dev)
    tmax = (torch.rand(n_q, n_b, B * T, generator=gen) * 5).to(dev)
    lens_q = torch.randint(1, T + 1, (n_q, n_b, B), generator=gen).to(dev)
    flags = torch.tensor([1, 0, 1], dtype=torch.int32, device=dev)
    for lengths in (lens_q, lens_q[0].contiguous(), None):
        cur = torch.zeros(n_b, n_q, 2, device=dev)
        ops.token_range_finalize_batched(tmin, tmax, n_q, n_b, B, T, lengths, flags, 0.8, cur)
        for qi in range(n_q):
            for bi in range(n_b):
                L = None if lengths is None else (lengths[qi, bi] if lengths.dim() == 3 else lengths[bi])
                one = torch.zeros(2, device=dev)
                ops.token_range_finalize(tmin[qi, bi], tmax[qi, bi], B, T, L, bool(flags[qi].item()), 0.8, ops.UPDATE_NONE, 0,
                                         None, None, 0, 63, False, None, one)
                assert torch.equal(cur[bi, qi], one), (qi, bi)
    with pytest.raises(ValueError):
        ops.token_range_finalize_batched(tmin, tmax, n_q, n_b, B, T, lens_q[:, :2].contiguous(), flags, 0.8, cur)
    with pytest.raises(ValueError):
        ops.token_minmax(torch.randn(B, T, 32, device=dev), 1, lens_q[0, 0], out=(tmin[0, 0, :16], tmax[0, 0, :16]))


def test_head_layout_fake_quant_helpers(dev):
    """merge_heads_fake_quant / split_heads_fake_quant (the context site quant_bert.py:184-188 and BART's head split
    quant_bart.py:226-243): in the plain quantising state the strided kernel writes the target layout itself; bit-equal
    to copy + fake-quant, for fixed (int32 zero-point) and learnable (fp32, repaired in the launch) quantizers, head sizes
    64 / 16 / 12 (the last one not a multiple of 16 bytes per row piece times four: general path) and an odd token count;
    with the observer on the two-step form runs and the statistics advance."""
    from types import SimpleNamespace as NS
    from outlier_suppression_amd.quantization import Quantizer
    from outlier_suppression_amd.util_layernorm import merge_heads_fake_quant, split_heads_fake_quant
    gen = torch.Generator().manual_seed(5)
    for kind in ("FixedFakeQuantize", "LSQPlusFakeQuantize"):
        for (b, h, t, d) in ((4, 12, 128, 64), (2, 4, 37, 16), (3, 2, 5, 12), (1, 16, 1024, 64)):
            q = Quantizer(None, NS(quantizer=kind, observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)).to(dev)
            ctx = (torch.randn(b, h, t, d, generator=gen) * 3).to(dev)
            lengths = torch.randint(1, t + 1, (b,), generator=gen).to(dev)
            q.enable_observer(); q.disable_fake_quant()
            with torch.no_grad():
                flat = ctx.permute(0, 2, 1, 3).contiguous().view(b, t, h * d)
                assert merge_heads_fake_quant(q, ctx, lengths).data_ptr() != ctx.data_ptr()
                cnt = q.observer.cnt
                q.disable_observer(); q.enable_fake_quant()
                if kind == "LSQPlusFakeQuantize":
                    q.scale.data.neg_()                               # the launch repairs it (fake_quant.py:188-191)
                ref = q(flat.clone(), lengths, 1)
                scale_after = q.scale.detach().clone()
                if kind == "LSQPlusFakeQuantize":
                    q.scale.data.neg_()
                y = merge_heads_fake_quant(q, ctx, lengths)
                assert y.shape == (b, t, h * d) and y.is_contiguous() and torch.equal(y, ref)
                assert torch.equal(q.scale.detach(), scale_after)
                heads = split_heads_fake_quant(q, flat, h, lengths)
                assert heads.shape == (b, h, t, d) and heads.is_contiguous()
                assert torch.equal(heads, ref.view(b, t, h, d).transpose(1, 2))
                assert q.observer.cnt == cnt
                q.enable_observer()
                y2 = merge_heads_fake_quant(q, ctx, lengths)          # both flags on: the general path, statistics advance
                assert q.observer.cnt == cnt + 1 and y2.shape == (b, t, h * d)
            # under autograd the two-step form runs and gradients reach the parameters
            q.disable_observer()
            if kind == "LSQPlusFakeQuantize":
                out = merge_heads_fake_quant(q, ctx, lengths)
                out.sum().backward()
                assert q.scale.grad is not None


def test_reciprocal_division_of_the_mse_grid_is_the_ieee_division(dev):
    """The all-candidates launch of the MSE grid (csrc/observers_extra.hip) forms x / scale as y = RN(1 / s), q0 = x * y and
    two fma corrections instead of dividing -- the correctly rounded quotient for every admitted operand pair (Markstein's
    theorem; guards: significand of s not all ones, |x| and s within [2^-60, 2^60]).  osq_selftest_division counts the
    pairs on which the two differ: 2^26 random pairs over the whole admitted exponent range, plus divisors with
    awkward significands (all ones but the last bit, 1.0, 1 + ulp) against dividends round the rounding boundaries."""
    import ctypes
    from outlier_suppression_amd import _hip
    lib = _hip.load()
    gen = torch.Generator(device=dev).manual_seed(77)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    n = 1 << 24
    for rep in range(4):
        mant_x = torch.rand(n, device=dev, generator=gen) + 1.0
        mant_s = torch.rand(n, device=dev, generator=gen) + 1.0
        ex = torch.randint(-59, 60, (n,), device=dev, generator=gen).float()
        es = torch.randint(-26, 60, (n,), device=dev, generator=gen).float()        # scales are floored at 1e-8 by calculate_qparams
        sign = torch.where(torch.rand(n, device=dev, generator=gen) < 0.5, -1.0, 1.0)
        x = sign * mant_x * torch.exp2(ex)
        s = mant_s * torch.exp2(es)
        if rep == 3:          # awkward divisors, dividends built as k * s (+- an ulp): quotients on rounding boundaries
            bits = torch.randint(0, 3, (n,), device=dev, generator=gen)
            pat = torch.tensor([0x3fffffe0 | 0x1e, 0x3f800000, 0x3f800001], dtype=torch.int32, device=dev)[bits]
            s = pat.view(torch.float32) * torch.exp2(torch.randint(-20, 10, (n,), device=dev, generator=gen).float())
            k = torch.randint(-64, 64, (n,), device=dev, generator=gen).float() + 0.5
            x = (k * s)
            x = (x.view(torch.int32) + torch.randint(-2, 3, (n,), device=dev, generator=gen).int()).view(torch.float32)
        _hip.check(lib.osq_selftest_division(_hip.ptr(x), _hip.ptr(s), n, _hip.ptr(bad), _hip.stream_ptr(dev)), "selftest_division")
    assert int(bad.item()) == 0


def test_qkv_headsplit_sites_in_one_launch(dev):
    """ops.fake_quant_headsplit_multi / util_layernorm.qkv_heads_fake_quant: the query / key / value activation quantizers of a
    self-attention block (quant_bert.py:148-155) as ONE launch -- torch.equal to the three per-site launches, parameter
    repair of the learnable quantizers included (fake_quant.py:188-191), for Fixed / LSQ+ quantizers and BERT / BART
    geometries; any state other than plain quantising hands the sites back to the per-site path."""
    from outlier_suppression_amd import util_layernorm as UL
    from outlier_suppression_amd.quantization import Quantizer
    gen = torch.Generator().manual_seed(77)
    for quantizer in ("LSQPlusFakeQuantize", "FixedFakeQuantize", "LSQFakeQuantize"):
        for (B, T_, h, d) in ((32, 128, 12, 64), (4, 33, 3, 16), (2, 7, 16, 64), (3, 5, 2, 4)):
            cfg = NS(quantizer=quantizer, observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
            qs = [Quantizer(None, cfg).to(dev) for _ in range(3)]
            xs = [(torch.randn(B, T_, h * d, generator=gen) * (1.0 + i)).to(dev) for i in range(3)]
            for i, q in enumerate(qs):
                q.enable_fake_quant()
                q.scale.data.fill_(0.05 * (i + 1) * (-1.0 if quantizer != "FixedFakeQuantize" and i == 1 else 1.0))   # a negative scale: the repair must run
                q.zero_point.data.fill_(20 + 5 * i)
            with torch.no_grad():
                saved = [(q.scale.data.clone(), q.zero_point.data.clone()) for q in qs]
                fused = UL.qkv_heads_fake_quant(qs, xs, h)
                assert fused is not None, (quantizer, B, T_, h, d)
                after = [(q.scale.data.clone(), q.zero_point.data.clone()) for q in qs]
                for q, (s0, z0) in zip(qs, saved):
                    q.scale.data.copy_(s0); q.zero_point.data.copy_(z0)
                for i, (q, x) in enumerate(zip(qs, xs)):
                    want = q(x.view(B, T_, h, d).permute(0, 2, 1, 3))
                    assert fused[i].shape == (B, h, T_, d) and fused[i].is_contiguous()
                    assert torch.equal(fused[i], want), (quantizer, i, B, T_, h, d)
                    assert torch.equal(q.scale.data, after[i][0]) and torch.equal(q.zero_point.data, after[i][1])
                    if quantizer != "FixedFakeQuantize":
                        assert q.scale.item() > 0
    # states the one launch does not take: observing, gradients wanted, another shape
    cfg = NS(quantizer="LSQPlusFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    qs = [Quantizer(None, cfg).to(dev) for _ in range(3)]
    xs = [torch.randn(2, 8, 64, device=dev) for _ in range(3)]
    for q in qs:
        q.enable_fake_quant()
    assert UL.qkv_heads_fake_quant(qs, xs, 4) is None                     # autograd is on: learn-scale keeps the per-site path
    with torch.no_grad():
        assert UL.qkv_heads_fake_quant(qs, xs, 4) is not None
        qs[1].enable_observer()
        assert UL.qkv_heads_fake_quant(qs, xs, 4) is None
        qs[1].disable_observer()
        assert UL.qkv_heads_fake_quant(qs, [xs[0], xs[1][:, :4], xs[2]], 4) is None
        UL.FUSE_QKV = False
        try:
            assert UL.qkv_heads_fake_quant(qs, xs, 4) is None
        finally:
            UL.FUSE_QKV = True
