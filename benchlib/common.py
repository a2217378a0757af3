"""Shared pieces of the benchmark: the BASELINE tensor, the quantizer of the timed step, constants."""
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPE = (256, 128, 768)          # BASELINE.json: BERT-base 256 x 128 x 768 activations
PERCENTILE = 0.95
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
COPY_RATE_GBS = 6290.0   # float4 copy kernel on MI355X (MI355X_MICROARCH.md): the practical ceiling of a read + write stream
GIB = float(1 << 30)


def make_inputs(dev, n_buffers, seed):
    """BASELINE.md section 4 synthetic inputs: randn with 6 seeded outlier hidden dims x20; lengths randint(8,129)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    outliers = torch.randperm(SHAPE[2], generator=g)[:6]
    lengths = torch.randint(8, 129, (SHAPE[0],), generator=g)
    gd = torch.Generator(device=dev).manual_seed(seed)
    xs = []
    for _ in range(n_buffers):
        x = torch.randn(*SHAPE, device=dev, generator=gd)
        x[..., outliers.to(dev)] *= 20.0
        xs.append(x)
    return xs, lengths


def _ops_order():
    from outlier_suppression_amd import ops
    return ops.reference_sum_order("mse")


def make_quantizer(dev):
    from types import SimpleNamespace as NS
    from outlier_suppression_amd.quantization import Quantizer
    cfg = NS(quantizer="LSQPlusFakeQuantize", observer="AvgPruneMinMaxObserver", bit=6, symmetric=False, ch_axis=-1)
    q = Quantizer(None, cfg).to(dev)
    q.observer.set_name("bert.encoder.layer.0.output.LayerNorm.layernorm_post_act_fake_quantize.observer")
    q.observer.set_percentile(PERCENTILE)
    q.enable_observer()
    q.enable_fake_quant()
    return q



# OSQ_BENCH_SHORT=1: the calibration flows with 2 batches, 3 candidates and 1 learn-scale epoch, run once -- the same kernels
# in the same states, a few thousand dispatches instead of a few hundred thousand: what the PMC passes of
# tools/collect_calibration_profiles.sh profile (rocprofv3 --pmc costs milliseconds per dispatch).  Never a measured wall-clock.
SHORT = os.environ.get("OSQ_BENCH_SHORT") == "1"



def device_identity(dev):
    """Something that tells two physical GPUs apart: the device's UUID, else its PCI address."""
    p = torch.cuda.get_device_properties(dev)
    uuid = getattr(p, "uuid", None)
    pci = ":".join(str(getattr(p, k, "?")) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    return f"{uuid}|{pci}|{p.name}"

