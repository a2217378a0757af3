"""CPU-only: the C-ABI library loads and exports every symbol include/osq_hip.h declares;
host-side logic (factory, flags, name-substring togglers, state-dict keys, fail-loudly rules)."""
import ctypes
import os
import re
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "osq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(osq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from outlier_suppression_amd import _hip
    lib = _hip.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/osq_hip.h but not exported"
        assert name in _hip.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_hip.SIGNATURES) == declared
    assert lib.osq_abi_version() >= 1
    assert lib.osq_workspace_bytes() >= 4096


def test_argument_validation_without_gpu():
    """Invalid arguments are rejected before any launch, so this is safe on a CPU box."""
    from outlier_suppression_amd import _hip
    lib = _hip.load()
    rc = lib.osq_fake_quant_per_tensor(None, None, None, 16, None, None, 0, 0, 1.0, 0, 63, None)
    assert rc == -1 and b"null" in lib.osq_last_error()
    rc = lib.osq_calculate_qparams(None, None, 4, 0, 63, 0, None, None, 0, None)
    assert rc == -1
    rc = lib.osq_token_range_finalize(ctypes.c_void_p(16), ctypes.c_void_p(16), 2, 2, None, 1, 1.5, 0, 0, None, None, None,
                                      0, 63, 0, None, None, 0, None, None, None)
    assert rc == -1 and b"percentile" in lib.osq_last_error()


SHIPPED_SWITCHES = {"mse_sum_order": (0, 8), "mse_rows_order": (8, 8), "fused_step": (0, 1), "mse_resident": (0, 1), "final_fast": (0, 1),
                    "select_shortcut": (0, 1), "mse_memo": (0, 1), "fused_spin_limit": (5, 0), "mse_spin_limit": (5, 0)}
AB_KNOBS = ("fq_unroll", "fq_max_blocks", "fq_nt", "fq_headsplit", "stream_wt", "bwd_blocks", "bwd_order_chunks", "ln_blocks",
            "obs_blocks", "tok_nt", "fused_gate", "fused_grid", "select_hint", "mse_round_groups", "mse_lean", "mse_grid_all", "mse_dbg")


def test_release_library_accepts_only_the_shipped_switches():
    """osq_set_tuning of the RELEASE library: the summation-order, path-selection and wait-bound switches only; the
    performance A/B knobs are compile-time constants there (their keys are refused) and variables only in the
    -DOSQ_TUNABLE build (`make dbg`), which osq_build_flags() identifies."""
    from outlier_suppression_amd import _hip
    lib = _hip.load()
    if os.environ.get("OSQ_HIP_LIBRARY"):
        pytest.skip("another build was asked for through OSQ_HIP_LIBRARY")
    assert lib.osq_build_flags() == 0, "the in-tree libosq_hip.so must be the release build"
    prev = {}
    from outlier_suppression_amd import ops
    for key, (probe, back) in SHIPPED_SWITCHES.items():
        assert lib.osq_set_tuning(key.encode(), probe) == 0, (key, lib.osq_last_error())
        prev[key] = ops._tuning.get(key, back)
        assert lib.osq_set_tuning(key.encode(), prev[key]) == 0, key
    for key in AB_KNOBS:
        assert lib.osq_set_tuning(key.encode(), 1) == -1, f"{key}: an A/B knob must not be settable in the release library"
        assert b"unknown key" in lib.osq_last_error()
    assert lib.osq_set_tuning(b"no_such_key", 1) == -1
    # what the header documents is what the library does
    header = open(os.path.join(ROOT, "include", "osq_hip.h")).read()
    for key in list(SHIPPED_SWITCHES) + [k for k in AB_KNOBS if k != "mse_dbg"]:
        assert f'"{key}"' in header, f"{key} is not documented at osq_set_tuning in include/osq_hip.h"


def test_per_call_escape_from_persistent_launches_is_part_of_the_module_api():
    """QuantizeBase.forward(..., persistent=False) / the `persistent` attribute / ops.observe_tokens_fake_quant(persistent=)
    reach OSQ_PARAM_NO_PERSISTENT (include/osq_hip.h): a multi-stream caller keeps one call off the whole-GPU launch without a
    process-global switch."""
    import inspect
    from outlier_suppression_amd import _hip, ops
    from outlier_suppression_amd.quantization.fake_quant import FixedFakeQuantize, LSQPlusFakeQuantize, QuantizeBase
    header = open(os.path.join(ROOT, "include", "osq_hip.h")).read()
    assert re.search(r"OSQ_PARAM_NO_PERSISTENT\s*=\s*32", header) and _hip.PARAM_NO_PERSISTENT == 32
    assert QuantizeBase.persistent is True
    for cls in (FixedFakeQuantize, LSQPlusFakeQuantize):
        sig = inspect.signature(cls.forward)
        assert list(sig.parameters)[:4] == ["self", "X", "observation_mask", "seq_pos"]          # the reference's signature first
        assert sig.parameters["persistent"].default is None
    assert inspect.signature(ops.observe_tokens_fake_quant).parameters["persistent"].default is True


def test_no_cpu_fallback():
    from outlier_suppression_amd import ops
    from outlier_suppression_amd.quantization import Quantizer
    cfg = NS(quantizer="FixedFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    q = Quantizer(None, cfg)
    x = torch.randn(2, 4, 8)
    assert q(x) is x                      # both switches off: identity, no compute
    q.enable_observer()
    with pytest.raises(RuntimeError, match="HIP device"):
        q(x)
    q.disable_observer()
    q.enable_fake_quant()
    with pytest.raises(RuntimeError, match="HIP device"):
        q(x)
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.calculate_qparams(torch.zeros(3), torch.ones(3), 0, 63, False)


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import outlier_suppression_amd.quantization, "
            "outlier_suppression_amd.ops, outlier_suppression_amd.gamma_migration, outlier_suppression_amd.calibration; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "outlier_suppression_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_factory_and_state_dict_keys(golden):
    from outlier_suppression_amd.quantization import Quantizer
    from outlier_suppression_amd.quantization.fake_quant import (FixedFakeQuantize, LSQFakeQuantize,
                                                                 LSQPlusFakeQuantize, QuantizeBase)
    from outlier_suppression_amd.quantization.quantized_module import (QLinear, QEmbedding, QConv2d, ObserverDict,
                                                                       FakeQuantizeDict)
    g = golden("modules")
    for k in range(int(g["n"])):
        quantizer, observer, bit, sym, ch_axis, kind, sdt, zdt = (str(v) for v in g[f"c{k}_info"])
        cfg = NS(quantizer=quantizer, observer=observer, bit=int(bit), symmetric=bool(int(sym)), ch_axis=int(ch_axis))
        q = Quantizer(None, cfg)
        assert isinstance(q, QuantizeBase) and type(q).__name__ == quantizer and type(q.observer).__name__ == observer
        assert sorted(q.state_dict().keys()) == [str(s) for s in g[f"c{k}_sdkeys"]]
        assert str(q.scale.dtype) == sdt and str(q.zero_point.dtype) == zdt
        assert (q.observer_enabled, q.fake_quant_enabled) == (0, 0)
    wcfg = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    lin = torch.nn.Linear(8, 5)
    ql = Quantizer(lin, wcfg)
    assert isinstance(ql, QLinear) and torch.equal(ql.weight, lin.weight) and ql.weight is not lin.weight
    assert isinstance(ql.weight_fake_quant, FixedFakeQuantize) and (ql.quant_min if hasattr(ql, "quant_min") else True)
    assert ql.weight_fake_quant.quant_min == -32 and ql.weight_fake_quant.quant_max == 31
    emb = Quantizer(torch.nn.Embedding(11, 4, padding_idx=0), wcfg)
    assert isinstance(emb, QEmbedding) and emb.padding_idx == 0
    conv = Quantizer(torch.nn.Conv2d(3, 4, 3, padding=1), wcfg)
    assert isinstance(conv, QConv2d)
    ln = torch.nn.LayerNorm(4)
    assert Quantizer(ln, wcfg) is ln
    assert {"MinMaxObserver", "AvgMinMaxObserver", "AvgPruneMinMaxObserver"} <= set(ObserverDict)
    assert set(FakeQuantizeDict) == {"FixedFakeQuantize", "LSQFakeQuantize", "LSQPlusFakeQuantize"}
    assert issubclass(LSQPlusFakeQuantize, QuantizeBase) and issubclass(LSQFakeQuantize, QuantizeBase)
    # with both switches off a Q-operator is the plain operator
    x = torch.randn(2, 8)
    assert torch.equal(ql(x), lin(x))


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        from outlier_suppression_amd.quantization import Quantizer
        a = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
        w = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
        self.dense = Quantizer(torch.nn.Linear(4, 4), w)
        self.attention_probs_post_act_fake_quantize = Quantizer(None, a)
        self.out_post_act_fake_quantize = Quantizer(None, NS(**{**vars(a), "quantizer": "FixedFakeQuantize"}))


def _flags(m):
    from outlier_suppression_amd.quantization.fake_quant import QuantizeBase
    return {n: (q.observer_enabled, q.fake_quant_enabled) for n, q in m.named_modules() if isinstance(q, QuantizeBase)}


def test_state_togglers_follow_reference_rules():
    from outlier_suppression_amd.quantization import (enable_calibration_woquantization, enable_calibration_quantization,
                                                      enable_quantization, disable_all)
    from outlier_suppression_amd.quantization.state import set_observer_name
    m = _Toy()
    W, A1, A2 = "dense.weight_fake_quant", "attention_probs_post_act_fake_quantize", "out_post_act_fake_quantize"
    enable_calibration_woquantization(m, quantizer_type="weight_fake_quant")
    assert _flags(m) == {W: (1, 0), A1: (0, 0), A2: (0, 0)}
    enable_calibration_woquantization(m, quantizer_type="act_fake_quant")
    assert _flags(m) == {W: (0, 0), A1: (1, 0), A2: (1, 0)}
    enable_calibration_quantization(m, quantizer_type="fake_quant")      # learnable ones keep the observer off
    assert _flags(m) == {W: (1, 1), A1: (0, 1), A2: (1, 1)}
    enable_quantization(m, except_quantizer=[A2])
    assert _flags(m) == {W: (0, 1), A1: (0, 1), A2: (0, 0)}
    disable_all(m)
    assert set(_flags(m).values()) == {(0, 0)}
    set_observer_name(m)
    assert m.attention_probs_post_act_fake_quantize.observer.name == A1 + ".observer"
    assert m.dense.weight_fake_quant.observer.name == W + ".observer"


def test_state_dict_round_trip_with_grown_buffers():
    """fake_quant.py:66-97: per-channel scale / zero_point change size after the first observation;
    loading must accept the stored shapes."""
    from outlier_suppression_amd.quantization import Quantizer
    w = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    src = Quantizer(torch.nn.Linear(4, 6), w)
    fq = src.weight_fake_quant
    fq.scale = torch.rand(6)
    fq.zero_point = torch.zeros(6, dtype=torch.int32)
    fq.observer.min_val = -torch.rand(6)
    fq.observer.max_val = torch.rand(6)
    sd = src.state_dict()
    dst = Quantizer(torch.nn.Linear(4, 6), w)
    dst.load_state_dict(sd)
    assert torch.equal(dst.weight_fake_quant.scale, fq.scale)
    assert torch.equal(dst.weight_fake_quant.observer.max_val, fq.observer.max_val)
    a = NS(quantizer="LSQPlusFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    q1, q2 = Quantizer(None, a), Quantizer(None, a)
    q1.scale.data.fill_(0.37)
    q1.zero_point.data.fill_(12.0)
    q2.load_state_dict(q1.state_dict())
    assert q2.scale.item() == pytest.approx(0.37) and q2.zero_point.item() == 12.0


def test_token_view_matches_reference_permutation():
    """ops.token_view describes the tensor observer.py:72-80 would build by permute+reshape."""
    from outlier_suppression_amd import ops
    from oracle.observer_oracle import _tokens_first
    rng = np.random.default_rng(0)
    cases = [((3, 5, 8), 1), ((3, 8, 5), 2), ((2, 3, 5, 4), 2), ((2, 3, 4, 5), 3), ((2, 5, 3, 4), 1)]
    for shape, sp in cases:
        x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
        for xv in (x, x.transpose(-1, -2).contiguous().transpose(-1, -2)):
            v = ops.token_view(xv, sp)
            want = _tokens_first(xv.numpy(), sp)
            flat = xv.contiguous().view(-1) if xv.is_contiguous() else None
            base = xv.numpy()
            storage = np.lib.stride_tricks.as_strided(base, shape=(v.batch, v.tokens, v.feat_outer, v.feat_inner),
                                                      strides=[4 * s for s in (v.stride_batch, v.stride_token,
                                                                               v.stride_outer, v.stride_inner)])
            assert np.array_equal(storage.reshape(v.batch, v.tokens, -1), want)
    # BART quirk: [B*h, T, S] with a length-B mask only covers the first B rows (observer.py:82)
    v = ops.token_view(torch.zeros(8, 5, 5), 1, n_lengths=2)
    assert v.batch == 2


def test_quantize_model_takes_the_reference_config_form():
    """quant_model.quantize_model(fp_model, config) -- the reference's call (solver/quant_model.py:31-50) with a parsed
    config (nested dicts as yaml.safe_load gives them, or attribute namespaces): defaults filled in, model_type / task_type
    written back, same wrapper as the explicit form."""
    from types import SimpleNamespace as NS
    import transformers as T
    from outlier_suppression_amd.quant_model import quantize_model, get_model_task_type
    from outlier_suppression_amd.model.quant_bert import QuantizedBertForQuestionAnswering
    from outlier_suppression_amd.quantization.fake_quant import LSQPlusFakeQuantize, FixedFakeQuantize
    cfg = T.BertConfig(vocab_size=50, hidden_size=16, num_hidden_layers=1, num_attention_heads=2, intermediate_size=32,
                       max_position_embeddings=20)
    fp = T.BertForQuestionAnswering(cfg)
    a_q = dict(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    w_q = dict(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
    config = {"quant": {"a_qconfig": a_q, "w_qconfig": w_q, "calibrate": 256}, "model": {},
              "data": NS(dataset_name="squad_v2")}
    model = quantize_model(fp, config)
    assert isinstance(model, QuantizedBertForQuestionAnswering) and model.is_remove_padding is True
    assert config["model"] == {"model_type": "bert", "task_type": "squad_v2"}
    assert config["quant"]["backend"] == "academic" and config["quant"]["ln"]["delay"] is False
    acts = [m for n, m in model.named_modules() if n.endswith("post_act_fake_quantize")]
    weights = [m for n, m in model.named_modules() if n.endswith("weight_fake_quant")]
    assert acts and all(isinstance(m, LSQPlusFakeQuantize) and m.bit == 6 for m in acts)
    assert weights and all(isinstance(m, FixedFakeQuantize) and m.ch_axis == 0 for m in weights)
    ns = NS(quant=NS(a_qconfig=NS(**a_q), w_qconfig=NS(**w_q), ln=NS(delay=True)), model=NS(), data=NS(dataset_name="squad"))
    model2 = quantize_model(fp, ns)
    assert [n for n, _ in model2.named_modules()] == [n for n, _ in model.named_modules()]
    assert ns.model.task_type == "squad" and ns.quant.ln.delay is True
    assert get_model_task_type("robertaforsequenceclassification", NS(dataset_name="mnli")) == ("glue", "roberta")
    assert get_model_task_type("bartforconditionalgeneration", NS(dataset_name="xsum")) == ("summ", "bart")
    with pytest.raises(NotImplementedError):
        get_model_task_type("bertforsequenceclassification", NS(dataset_name="imdb"))


def test_resident_kernels_do_not_spill():
    """The resident MSEFast kernels exist to keep a tensor in registers across hundreds of loss evaluations: a spilled
    VGPR is a scratch round trip in every evaluation.  Round 2 shipped them with 102-486 spilled VGPRs; this reads the
    code-object metadata of the built library (no GPU needed) and fails on the first spilled or scratch-backed one.
    The streaming path of the fused observe + fake-quant kernel is held to the same bar; its two selector workgroups call
    a noinline selection (stack for callee-saved registers) and park ONE register of SGPR copies across that call."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources
    rows = kernel_resources.kernel_resources()
    assert len(rows) > 50, "could not read the code-object notes of libosq_hip.so"
    resident = [r for r in rows if "msefast_resident" in r["name"]]
    assert len(resident) >= 11
    for r in resident:
        assert r.get("vgpr_spill_count", 0) == 0 and r.get("private_segment_fixed_size", 0) == 0, r
        assert r.get("max_flat_workgroup_size") == 512, r
    for r in rows:
        if "msefast_rows_kernel" in r["name"]:
            assert r.get("vgpr_spill_count", 0) == 0, r
        if "observe_fq_fused_kernel" in r["name"]:
            assert r.get("vgpr_spill_count", 0) <= 1, r


def test_strict_and_fast_switches_from_the_environment():
    """The switches applied when the library is first loaded (outlier_suppression_amd.reset_tier): by default the MSEFast
    sums follow the reference's one-thread order (8-lane host) and the LSQ / LSQ+ backward's sums are order-free (round 5:
    ADVICE r04 -- the strict backward is 1.3x slower for parity with a patched-CPU fixture no upstream run has);
    OSQ_STRICT=1 puts both in the reference's order, OSQ_STRICT=0 neither; OSQ_FAST sets the one-launch LayerNorm site."""
    import subprocess
    import sys
    code = ("from outlier_suppression_amd import _hip, ops, util_layernorm as UL; _hip.load(); "
            "print(ops.reference_sum_order('mse'), ops.reference_sum_order('bwd'), UL.FUSE_LAYERNORM, UL.FUSE_ACTIVATION)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("OSQ_STRICT", "OSQ_FAST", "OSQ_STRICT_SIMD")}
    run = lambda **kw: subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(env, **kw), capture_output=True, text=True, timeout=300)
    from outlier_suppression_amd import util_layernorm as UL
    ln = str(UL.FUSE_LAYERNORM)
    out = run()
    assert out.stdout.split() == ["8", "0", ln, "True"], out.stdout + out.stderr
    out = run(OSQ_STRICT="0", OSQ_FAST="1")
    assert out.stdout.split() == ["0", "0", "True", "True"], out.stdout + out.stderr
    out = run(OSQ_STRICT="1", OSQ_FAST="0")
    assert out.stdout.split() == ["8", "8", "False", "True"], out.stdout + out.stderr
    out = run(OSQ_STRICT_SIMD="16")
    assert out.stdout.split() == ["16", "0", ln, "True"], out.stdout + out.stderr
    out = run(OSQ_STRICT="1", OSQ_STRICT_SIMD="16")
    assert out.stdout.split() == ["16", "16", ln, "True"], out.stdout + out.stderr


def test_stale_library_is_refused():
    """_hip.load() compares osq_abi_version() with the number the Python host was written against (a stale
    libosq_hip.so used to be caught only when a new symbol happened to be missing)."""
    import re
    from outlier_suppression_amd import _hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "osq_hip.h")).read()
    assert int(re.search(r"#define OSQ_ABI_VERSION (\d+)", header).group(1)) == _hip.ABI_VERSION
    assert _hip.load().osq_abi_version() == _hip.ABI_VERSION
    import subprocess
    import sys
    code = ("from outlier_suppression_amd import _hip; _hip.ABI_VERSION += 1\n"
            "try:\n    _hip.load()\nexcept _hip.HipLibraryMissing as e:\n    print('refused:', 'rebuild' in str(e))")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == "refused: True", out.stdout + out.stderr


def test_ordered_sum_capacity_rule():
    """ops.ordered_sum_fits mirrors the kernels' bound (csrc/aten_order.h: cascade step 2^P, P = max(4, ceil_log2(rows) / 4) <= 5,
    rows = n / lanes / 4): up to 2^23 rows -- 268 M fp32 elements on 8 lanes, 134 M float64 elements on 4."""
    from outlier_suppression_amd import ops
    assert ops.ordered_sum_fits(5, 8) and ops.ordered_sum_fits(25165824, 8) and ops.ordered_sum_fits(1 << 28, 8)
    assert not ops.ordered_sum_fits((1 << 28) + 32, 8)
    assert ops.ordered_sum_fits(1 << 27, 4) and not ops.ordered_sum_fits((1 << 27) + 16, 4)
    assert ops.ordered_sum_fits(1 << 29, 16) and not ops.ordered_sum_fits((1 << 29) + 64, 16)
