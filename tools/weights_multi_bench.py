"""Development aid: the one-launch weight fake-quant of BERT-base (77 tensors, 110 M parameters) -- kernel time by stream
events round the bare library call (AB_LIB selects another build), result compared with the per-operator launches."""
import os, sys
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers as T
from outlier_suppression_amd import _hip
if os.environ.get("AB_LIB"):
    _hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), os.environ["AB_LIB"])
from outlier_suppression_amd.quant_model import quantize_model
from outlier_suppression_amd.quantization import enable_calibration_woquantization, enable_quantization
from outlier_suppression_amd.quantization import weight_cache as WC
dev = torch.device("cuda:0")
a_q = NS(quantizer="FixedFakeQuantize", observer="AvgMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
fp = T.BertForSequenceClassification(T.BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).eval().to(dev)
m = quantize_model(fp, w_q, a_q).to(dev)
ids = torch.randint(1000, 30000, (4, 16), device=dev)
with torch.no_grad():
    enable_calibration_woquantization(m, quantizer_type="weight_fake_quant"); m(input_ids=ids)
    enable_calibration_woquantization(m, quantizer_type="act_fake_quant"); m(input_ids=ids)
    enable_quantization(m)
    n = WC.prepare_weights(m)
    _, table, ends, views, total_rows = WC._PLAN[m]
    lib = _hip.load()
    ok = True
    for mod in m.modules():
        if "weight_fake_quant" in mod.__dict__.get("_modules", {}):
            ok &= torch.equal(WC._CACHE[mod][1], mod.weight_fake_quant(mod.weight))
    ts = []
    for i in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _hip.check(lib.osq_fake_quant_weights_multi(table.data_ptr(), ends.data_ptr(), n, total_rows, _hip.stream_ptr(dev)), "multi")
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    nbytes = sum(v.numel() for v in views) * 8
    print(f"{n} tensors, {total_rows} rows, {nbytes / 1e6:.0f} MB: median {ts[6]:.1f} us = {nbytes / ts[6] / 1e6:.2f} TB/s; equal to per-operator launches: {ok}")
