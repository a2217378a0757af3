"""Development aid: is a quantized forward host-bound?  Eager wall-clock per forward against the replay of the same forward
captured into a hipGraph (BASELINE configs 1 and 4: BERT-base [32,128], BART-base dims [4,1024] + [4,62])."""
import os, sys, time, logging
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers as T
from outlier_suppression_amd import token_wise_clipping as TWC
from outlier_suppression_amd.quant_model import quantize_model
from outlier_suppression_amd.quantization import enable_calibration_woquantization, enable_quantization, disable_all
from outlier_suppression_amd.quantization.state import set_observer_name
logging.getLogger("transformer").setLevel(logging.WARNING)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
a_q = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
w_q = NS(quantizer="FixedFakeQuantize", observer="MinMaxObserver", bit=6, symmetric=True, ch_axis=0)


def batch(B, Tn, vocab, lo):
    L = torch.randint(lo, Tn + 1, (B,), generator=g)
    mask = (torch.arange(Tn)[None, :] < L[:, None]).long()
    ids = torch.randint(1000, vocab - 1000, (B, Tn), generator=g) * mask + (1 - mask)
    return {"input_ids": ids.to(dev), "attention_mask": mask.to(dev)}


for which in sys.argv[1:] or ["bert", "bart"]:
    if which == "bert":
        fp = T.BertForSequenceClassification(T.BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).eval().to(dev)
        b = batch(32, 128, 30522, 8)
        b["token_type_ids"] = torch.zeros_like(b["input_ids"])
    else:
        cfg = T.BartConfig(d_model=768, encoder_layers=6, decoder_layers=6, encoder_attention_heads=12, decoder_attention_heads=12,
                           encoder_ffn_dim=3072, decoder_ffn_dim=3072, max_position_embeddings=1024, dropout=0.0,
                           attention_dropout=0.0, activation_dropout=0.0)
        fp = T.BartForConditionalGeneration(cfg).eval().to(dev)
        b = batch(4, 1024, 50265, 256)
        dm = (torch.arange(62)[None, :] < torch.tensor([62, 40, 30, 20])[:, None]).long()
        b["decoder_input_ids"] = (torch.randint(1000, 49000, (4, 62), generator=g) * dm + (1 - dm)).to(dev)
        b["decoder_attention_mask"] = dm.to(dev)
    m = quantize_model(fp, w_q, a_q).to(dev)
    set_observer_name(m)
    with torch.no_grad():
        enable_calibration_woquantization(m, quantizer_type="weight_fake_quant"); m(**b)
        TWC.set_ratio(m, 0.9)
        enable_calibration_woquantization(m, quantizer_type="act_fake_quant"); m(**b)
        enable_quantization(m)
        for _ in range(3):
            ref = m(**b)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = m(**b)[0]
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            out = m(**b)[0]
        host = (time.perf_counter() - t0) / 20
        torch.cuda.synchronize()
        print(f"{which}: eager {eager * 1e3:.2f} ms per forward, host enqueue {host * 1e3:.2f} ms", flush=True)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m(**b)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                gout = m(**b)[0]
            torch.cuda.synchronize()
            graph.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                graph.replay()
            torch.cuda.synchronize()
            rep = (time.perf_counter() - t0) / 20
            print(f"{which}: graph replay {rep * 1e3:.2f} ms per forward, bit-equal to eager: {torch.equal(gout, ref)}", flush=True)
        except Exception as e:
            print(f"{which}: capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
