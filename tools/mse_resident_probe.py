#!/usr/bin/env python3
"""Round latency of the RESIDENT reference-order MSEFast searches (csrc/msefast_resident_ordered.h): k sites of [32,128,768]
(unmasked, float64 call), one persistent launch; microseconds per loss evaluation of the slowest search = launch time / its
evaluation count.  Compared with the streaming rounds on the same sites."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import outlier_suppression_amd as osq  # noqa: E402
from outlier_suppression_amd import ops  # noqa: E402
from outlier_suppression_amd.quantization.deferred import deferred_observation  # noqa: E402
from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for ki in (8, 10):
    ops.set_tuning("mse_resident_items", ki)
    for shape, k in (((32, 128, 768), 1), ((32, 128, 768), 2), ((32, 128, 768), 4), ((8, 128, 768), 8), ((32, 128, 3072), 1)):
        xs = []
        for _ in range(k):
            x = torch.randn(*shape, generator=g)
            x[..., 3] *= 12
            xs.append(x.to(dev))
        for resident in (True, False):
            ops.ORDERED_RESIDENT = resident
            obs = [AvgMSEFastObserver(bit=6, symmetric=False).to(dev) for _ in range(k)]
            for ob in obs:
                object.__setattr__(ob, "_defer_ok", True)
            with deferred_observation() as sites:
                for ob, x in zip(obs, xs):
                    ob(x, None, 1)
                sites.flush()                      # fp32 call (streaming rounds either way)
                for ob, x in zip(obs, xs):
                    ob(x * 1.01, None, 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sites.flush()                      # float64 call
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            nfev = max(int(ob.last_nfev.max().item()) for ob in obs)
            print(f"items/wg {ki}  {k} x {list(shape)}  {'resident' if resident else 'rounds  '}: {dt * 1e3:8.2f} ms, max nfev {nfev}, {dt / nfev * 1e6:7.2f} us per evaluation")
ops.ORDERED_RESIDENT = True
