"""Development aid: do the quantized forwards of a grid-search candidate (8 independent batches) finish sooner when they
run on several streams at once?  BERT-base [32,128]; each batch's forward is captured into its own hipGraph and the eight
graphs are replayed on 1, 2, 4 streams (same kernels, same shapes: bit-identical results)."""
import os, sys, time, logging
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers as T
from outlier_suppression_amd import token_wise_clipping as TWC
from outlier_suppression_amd.quant_model import quantize_model
from outlier_suppression_amd.quantization import enable_calibration_woquantization, enable_quantization
from outlier_suppression_amd.quantization.state import set_observer_name
logging.getLogger("transformer").setLevel(logging.WARNING)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
a_q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)
Tn = int(os.environ.get("PROBE_T", "128"))


def batch(B, Tn, vocab, lo):
    L = torch.randint(lo, Tn + 1, (B,), generator=g)
    mask = (torch.arange(Tn)[None, :] < L[:, None]).long()
    ids = torch.randint(1000, vocab - 1000, (B, Tn), generator=g) * mask + (1 - mask)
    return {"input_ids": ids.to(dev), "attention_mask": mask.to(dev), "token_type_ids": torch.zeros(B, Tn, dtype=torch.long, device=dev)}


fp = T.BertForSequenceClassification(T.BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).eval().to(dev)
batches = [batch(32, Tn, 30522, 8) for _ in range(8)]
m = quantize_model(fp, w_q, a_q).to(dev)
set_observer_name(m)
with torch.no_grad():
    enable_calibration_woquantization(m, quantizer_type="weight_fake_quant"); m(**batches[0])
    TWC.set_ratio(m, 0.9)
    enable_calibration_woquantization(m, quantizer_type="act_fake_quant"); m(**batches[0])
    enable_quantization(m)
    for b in batches[:2]:
        m(**b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        outs = [m(**b)[0] for b in batches]
    torch.cuda.synchronize()
    print(f"eager, one stream: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per 8 forwards", flush=True)
    ref = [o.clone() for o in outs]
    graphs, gouts = [], []
    side = torch.cuda.Stream()
    for b in batches:
        gr = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(gr, stream=side):
            o = m(**b)[0]
        graphs.append(gr); gouts.append(o)
    torch.cuda.synchronize()
    for n_streams in (1, 2, 4, 8):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        for rep in range(4):
            if rep == 1:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            for i, gr in enumerate(graphs):
                with torch.cuda.stream(streams[i % n_streams]):
                    gr.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3 * 1e3
        same = all(torch.equal(a, b) for a, b in zip(ref, gouts))
        print(f"graphs on {n_streams} stream(s): {dt:.2f} ms per 8 forwards   bit-equal to eager: {same}", flush=True)
