"""CPU, build container only: the drop-in of INTEGRATION.md section 1, exercised.

BASELINE.json north_star: "keeping the repo's FakeQuantize/Observer nn.Module API so the quantized BERT/RoBERTa/BART
models in quant_transformer/model load unchanged".  tests/shim_probe.py imports the REFERENCE's model files
(model/quant_bert.py:598 QuantizedBertForSequenceClassification, :690 ...ForQuestionAnswering, quant_roberta.py,
quant_bart.py), its util_layernorm.py and its solver/gamma_migration.py UNMODIFIED, once on the reference's own
quantization package and once with ``quant_transformer.quantization`` aliased to ``outlier_suppression_amd.quantization``
in sys.modules, and describes what it built.  Everything a caller can see must be the same: module tree, quantizer names
and classes, state-dict keys / dtypes / shapes, the tree after the reference's delay_ln (which swaps LayerNorms and folds
gamma into this package's QLinear weights), a state-dict round trip, and -- all quantizers off -- the logits.
Each mode runs in its own interpreter: the two quantization packages must not meet in one sys.modules.
"""
import json
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not available")
HERE = os.path.dirname(os.path.abspath(__file__))


def _probe(mode):
    r = subprocess.run([sys.executable, os.path.join(HERE, "shim_probe.py"), mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("SHIM_JSON ")][-1]
    return json.loads(line[len("SHIM_JSON "):])


@pytest.fixture(scope="module")
def probes():
    return {m: _probe(m) for m in ("pure", "shim", "shim+ln")}


MODELS = ["bert-cls", "bert-qa", "roberta-cls", "roberta-qa", "bart"]


def test_reference_models_are_built_from_this_package(probes):
    assert probes["pure"]["Quantizer_from"].startswith("quant_transformer.")
    for mode in ("shim", "shim+ln"):
        assert probes[mode]["Quantizer_from"].startswith("outlier_suppression_amd.")
        assert probes[mode]["QuantizedModule_from"].startswith("outlier_suppression_amd.")
        for name in MODELS:
            assert probes[mode][name]["quantizer_package"] == ["outlier_suppression_amd"], (mode, name)
            assert len(probes[mode][name]["quantizers"]) >= 33


@pytest.mark.parametrize("name", MODELS)
def test_same_modules_quantizers_and_state_dict(probes, name):
    pure = probes["pure"][name]
    for mode in ("shim", "shim+ln"):
        got = probes[mode][name]
        assert got["modules"] == pure["modules"], mode
        assert got["quantizers"] == pure["quantizers"], mode
        assert got["quantizer_classes"] == pure["quantizer_classes"], mode
        assert got["state_dict"] == pure["state_dict"], mode       # keys in order, dtypes (int32 zero_point of Fixed), shapes
    # after the REFERENCE's gamma migration on this package's modules
    a, b = dict(probes["shim"][name]["after_delay_ln"]), dict(pure["after_delay_ln"])
    assert a.pop("quantizer_package") == ["outlier_suppression_amd"] and b.pop("quantizer_package") == ["quant_transformer"]
    assert a == b


@pytest.mark.parametrize("name", MODELS[:4])
def test_fp_logits_and_round_trip(probes, name):
    pure, shim = probes["pure"][name], probes["shim"][name]
    assert shim["fp_logits"] == pure["fp_logits"]                   # every quantizer off: bit-identical FP model
    assert probes["shim+ln"][name]["fp_logits"] == pure["fp_logits"]
    assert shim["fp_logits_after_delay_ln"] == pure["fp_logits_after_delay_ln"]
    assert shim["round_trip"] == [[], []] and pure["round_trip"] == [[], []]
