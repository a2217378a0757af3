"""Development aid: where does a learn-scale step spend its time?  (run under tools/prof.sh)"""
import os, sys, time
from types import SimpleNamespace as NS
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import BertConfig, BertForSequenceClassification
from outlier_suppression_amd import token_wise_clipping as TWC
from outlier_suppression_amd.gamma_migration import delay_ln
from outlier_suppression_amd.quant_model import quantize_model
from outlier_suppression_amd.quantization import enable_calibration_woquantization, disable_all
from outlier_suppression_amd.quantization.state import set_observer_name
dev = torch.device("cuda:0")
torch.manual_seed(0)
fp = BertForSequenceClassification(BertConfig(num_labels=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).eval().to(dev)
g = torch.Generator().manual_seed(42)
batches = []
for _ in range(4):
    L = torch.randint(8, 129, (32,), generator=g)
    mask = (torch.arange(128)[None, :] < L[:, None]).long()
    ids = torch.randint(1000, 30000, (32, 128), generator=g) * mask
    batches.append({"input_ids": ids.to(dev), "attention_mask": mask.to(dev), "token_type_ids": torch.zeros_like(ids).to(dev)})
a_q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
m = quantize_model(fp, w_q, a_q).to(dev)
with torch.no_grad():
    out = [m(**b)[0].detach() for b in batches]
m = delay_ln(m, NS(a_qconfig=a_q, w_qconfig=w_q), NS(model_type="bert", task_type="glue"))
TWC.task_type = "glue"
disable_all(m); set_observer_name(m)
TWC.set_ratio(m, 0.9); TWC.calibrate(m, batches)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    TWC.learn_scale(NS(model=m), batches, out, {"lr": 1e-5, "epoch": 3})
    torch.cuda.synchronize(); print("learn_scale 12 steps:", round(time.perf_counter() - t0, 3), "s")
