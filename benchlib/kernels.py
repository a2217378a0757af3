"""Per-kernel table: every kernel of the path on the BASELINE tensor and the site shapes (detail section `kernels`)."""
import time

import torch

from .common import SHAPE, PERCENTILE, HBM_PEAK_GBS, COPY_RATE_GBS, make_quantizer


def kernel_table(dev, xs, lengths, reps=20):
    """Every kernel of the path on the BASELINE tensor, one at a time, each launch timed by HIP events that ride
    on its own dispatch packet (osq_time_next_launch), inputs cycled through buffers larger than the Infinity
    Cache.  Algorithmic bytes per BASELINE.md: fake-quant 8 B/elem, observers 4 B per observed elem, LSQ+ backward
    12 B/elem; the selection kernel reads the per-token extrema (8 B per token slot) and is latency/issue bound."""
    import ctypes
    from outlier_suppression_amd import _hip, ops
    lib = _hip.load()
    n = xs[0].numel()
    valid = int(lengths.sum().item()) * SHAPE[2]
    full = torch.full_like(lengths, SHAPE[1])
    s = torch.tensor([0.7], device=dev)
    zf = torch.tensor([31.0], device=dev)
    mn = torch.tensor(float("inf"), device=dev)
    mx = torch.tensor(float("-inf"), device=dev)
    cur = torch.empty(2, device=dev)
    gy = torch.randn_like(xs[0])
    tok = ops.token_minmax(xs[0], 1, lengths)

    def timed(which, fn):
        out = []
        for i in range(reps + 3):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            _hip.check(lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)), "timing_events_create")
            lib.osq_time_next_launch(which, a, b)
            fn(i)
            us = ctypes.c_float()
            _hip.check(lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)), "timing_elapsed_us")
            lib.osq_timing_events_destroy(a, b)
            if i >= 3:
                out.append(us.value)
        return sum(out) / len(out)

    rows = {}
    # What this clock reads for a launch that moves (almost) nothing: a 4 KiB fake-quant.  The site-size rows below sit on
    # this floor (a [32,128,768] site is 6.6-25 MB: 1-4 us of HBM time): their bandwidth fractions say "too small a tensor for
    # one launch", `us_above_floor` says how much of the launch is the kernel's own
    tiny = torch.randn(1024, device=dev)
    with torch.no_grad():
        floor_us = timed(_hip.TIME_FAKE_QUANT, lambda i: ops.fake_quant_per_tensor(tiny, s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4))
    rows["launch floor (fake-quant of 4 KiB)"] = {"avg_us": round(floor_us, 2),
                                                  "bound": "launch: dispatch + one HBM round trip + completion, as the dispatch events see it"}

    def hbm_row(us, nbytes, resident=False, cycled=False):
        """A row priced against HBM -- unless its bytes cannot have come from HBM: a launch that re-reads the same tensor
        every time with a working set under the 256 MB Infinity Cache (`resident`: the site-size rows), or any row of
        less than 256 MB whose rate exceeds what a float4 copy reaches on this part (6.29 TB/s), is labelled
        cache-resident and carries no fraction of the HBM peak.  `cycled`: the launches walk through 4 x 96 MiB inputs (and as
        many outputs): a working set of 384 MiB and more, HBM whatever the rate."""
        gbps = nbytes / us / 1e3
        row = {"avg_us": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(gbps, 1)}
        if not cycled and nbytes < 256e6 and (resident or gbps > COPY_RATE_GBS):
            row["bound"] = "cache-resident (same < 256 MB working set every launch: Infinity Cache, not HBM) + launch floor"
        else:
            row["bound"] = "hbm"
            row["frac_of_8TBps"] = round(gbps / HBM_PEAK_GBS, 3)
        return row

    def add(name, us, nbytes, resident=False, cycled=None):
        if "token_select" in name:     # two workgroups per problem on one CU each: exact order statistics, not a stream
            rows[name] = {"avg_us": round(us, 2), "bound": "one CU per side: VALU issue + LDS atomic rate (not HBM)",
                          "token_slots_MB": round(nbytes / 1e6, 2), "us_above_floor": round(us - floor_us, 2)}
            return
        # site / weight rows launch on the SAME tensor every time: Infinity-Cache regime whenever it fits
        resident = resident or name.startswith(("site ", "weight "))
        cycled = (not resident) if cycled is None else cycled     # the BASELINE-tensor rows cycle xs[i % 4]
        rows[name] = dict(hbm_row(us, nbytes, resident, cycled), us_above_floor=round(us - floor_us, 2))

    with torch.no_grad():
        add("fake_quant_forward", timed(_hip.TIME_FAKE_QUANT, lambda i: ops.fake_quant_per_tensor(
            xs[i % len(xs)], s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)), 8 * n)
        add("observe_flat (MinMax / AvgMinMax, no mask)", timed(_hip.TIME_OBSERVE_FLAT, lambda i: ops.observe_flat(
            xs[i % len(xs)], ops.UPDATE_RUNNING, 0, mn, mx, 0, 63, False)), 4 * n)
        add("token_minmax, all tokens", timed(_hip.TIME_TOKEN_MINMAX, lambda i: ops.token_minmax(xs[i % len(xs)], 1, full)), 4 * n)
        add("token_minmax, bench lengths", timed(_hip.TIME_TOKEN_MINMAX, lambda i: ops.token_minmax(xs[i % len(xs)], 1, lengths)), 4 * valid)
        add("token_select p=0.95 (32768 slots)", timed(_hip.TIME_TOKEN_SELECT, lambda i: ops.token_range_finalize(
            tok[0], tok[1], tok[2], tok[3], tok[4], True, PERCENTILE, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur)),
            8 * SHAPE[0] * SHAPE[1])
        # BASELINE.md section 2's "observer forward" row: the north-star observer ALONE (AvgPruneMinMaxObserver, fake-quant off
        # -- the state of every observer pass of token-wise clipping) = the two launches above, back to back
        for tag_, mm, nb in (("bench lengths", "token_minmax, bench lengths", 4 * valid), ("all tokens", "token_minmax, all tokens", 4 * n)):
            us = rows[mm]["avg_us"] + rows["token_select p=0.95 (32768 slots)"]["avg_us"]
            rows[f"observer alone (AvgPruneMinMax p=0.95: token_minmax + token_select), {tag_}"] = {
                **hbm_row(us, nb, cycled=True), "bound": "hbm + one CU per side for the selection",
                "note": "sum of the two launches' own durations; the kernel boundary between them (~1.7 us) is not in it"}
        # the default backward adds the two parameter gradients in float64 and rounds once (order-free); set_strict(backward=True)
        # adds autograd's four fp32 sums in ATen's one-thread order (bit-equal to the reference's CPU run, 1.3x slower)
        prev_order = ops.reference_sum_order("bwd")
        ops.set_tuning("bwd_sum_order", 0)
        try:
            add("lsq_plus_backward (default: order-free parameter gradients)", timed(_hip.TIME_LSQ_BACKWARD, lambda i: ops.lsq_backward_per_tensor(
                xs[i % len(xs)], gy, s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)), 12 * n)
            ops.set_tuning("bwd_sum_order", 8)
            add("lsq_plus_backward (set_strict(backward=True): gradients summed in the reference's order)", timed(_hip.TIME_LSQ_BACKWARD, lambda i: ops.lsq_backward_per_tensor(
                xs[i % len(xs)], gy, s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)), 12 * n)
        finally:
            ops.set_tuning("bwd_sum_order", prev_order)
        # LayerNorm site of a quantized block: GammaResidual -> split LayerNorm -> + beta/gamma -> fake-quant, one launch
        gamma = torch.rand(SHAPE[2], device=dev) + 0.5
        shift = torch.randn(SHAPE[2], device=dev)
        quant = (s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
        add("residual+layernorm+fake_quant (one launch)", timed(_hip.TIME_LAYERNORM, lambda i: ops.residual_layernorm_fake_quant(
            xs[i % len(xs)], gy, gamma, None, shift, 1e-5, quant)), 12 * n)
        # the same site as the eager sequence (4 launches; stream-order events around the whole sequence)
        import torch.nn.functional as F
        ev = []
        for i in range(reps + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = ops.gamma_residual(xs[i % len(xs)], gy, gamma)
            r = F.layer_norm(r, (SHAPE[2],), None, None, 1e-5)
            r += shift
            r = ops.fake_quant_per_tensor(r, s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        seq_us = sum(a.elapsed_time(b) for a, b in ev[3:]) / reps * 1e3
        # BASELINE.md section 4 secondary slices: one calibration site of BERT-base at batch 32 (launch-latency regime)
        for shp, sp in (((32, 128, 768), 1), ((32, 12, 128, 128), 2), ((32, 128, 3072), 1), ((32, 384, 768), 1)):
            xsite = torch.randn(*shp, device=dev)
            lsite = torch.randint(8, shp[sp] + 1, (shp[0],), device=dev)
            vsite = int(lsite.sum().item()) * (xsite.numel() // shp[0] // shp[sp])
            tag = "x".join(str(d) for d in shp)
            add(f"site {tag}: fake_quant_forward", timed(_hip.TIME_FAKE_QUANT, lambda i: ops.fake_quant_per_tensor(
                xsite, s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)), 8 * xsite.numel())
            add(f"site {tag}: token_minmax (masked)", timed(_hip.TIME_TOKEN_MINMAX, lambda i: ops.token_minmax(xsite, sp, lsite)), 4 * vsite)
            tk = ops.token_minmax(xsite, sp, lsite)
            add(f"site {tag}: token_select p=0.95", timed(_hip.TIME_TOKEN_SELECT, lambda i: ops.token_range_finalize(
                tk[0], tk[1], tk[2], tk[3], tk[4], True, PERCENTILE, ops.UPDATE_NONE, 0, None, None, 0, 63, False, None, cur)),
                8 * shp[0] * shp[sp])
            # learn-scale's backward at site size, both summation tiers (12 B per element: x, grad_out in, dx out)
            gsite = torch.randn_like(xsite)
            prev_order = ops.reference_sum_order("bwd")
            try:
                for order, what in ((0, "default: order-free"), (8, "set_strict(backward=True): reference order")):
                    ops.set_tuning("bwd_sum_order", order)
                    add(f"site {tag}: lsq_plus_backward ({what})", timed(_hip.TIME_LSQ_BACKWARD, lambda i: ops.lsq_backward_per_tensor(
                        xsite, gsite, s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)), 12 * xsite.numel())
            finally:
                ops.set_tuning("bwd_sum_order", prev_order)
        # ---- rows of SURVEY.md section 8d that have no dispatch-attached timer: stream-order events around the call
        # (they include one kernel boundary, ~2 us)
        def ev_timed(fn, inner=1):
            ev = []
            for i in range(reps + 3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(inner):
                    fn(i)
                e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            return sum(a.elapsed_time(b) for a, b in ev[3:]) / reps * 1e3 / inner

        def add_ev(name, us, nbytes, note=None, resident=False):
            rows[name] = dict(hbm_row(us, nbytes, resident, cycled=not resident), timer="stream events around the call (one kernel boundary included)")
            if note:
                rows[name]["note"] = note

        # the whole step as ONE launch, by mask
        from outlier_suppression_amd.quantization import Quantizer
        qf = make_quantizer(dev)
        for tag, lens in (("bench lengths", lengths), ("all tokens valid", full)):
            v = int(lens.sum().item()) * SHAPE[2]
            add(f"fused observe+fake-quant step, {tag}", timed(_hip.TIME_FUSED_STEP, lambda i: qf(xs[i % len(xs)], lens, 1)), 4 * v + 8 * n)
        # attention head-split views of [B,T,h,d] memory (quant_bert.py:128-150): q / v as [B,h,T,d], k as [B,h,d,T]
        mem = torch.randn(32, 128, 12, 64, device=dev)
        l32 = torch.randint(8, 129, (32,), device=dev)
        for tag, view, sp in (("32x12x128x64 (q/v view of [B,T,h,d])", mem.permute(0, 2, 1, 3), 2),
                              ("32x12x64x128 (key view, strided)", mem.permute(0, 2, 3, 1), 3)):
            add(f"site {tag}: fake_quant_forward", timed(_hip.TIME_FAKE_QUANT_STRIDED, lambda i: ops.fake_quant_per_tensor(
                view, s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4)), 8 * mem.numel())
            vv = int(l32.sum().item()) * 12 * 64
            add(f"site {tag}: token_minmax (masked)", timed(_hip.TIME_TOKEN_MINMAX, lambda i: ops.token_minmax(view, sp, l32)), 4 * vv)
        # weights: per-channel fake-quant (6-bit symmetric, ch_axis 0) and per-channel MinMax observer (+ qparams), one launch each
        for shp in ((768, 768), (3072, 768), (30522, 768)):
            w = torch.randn(*shp, device=dev) * 0.05
            ws_, wz_ = torch.full((shp[0],), 0.01, device=dev), torch.zeros(shp[0], dtype=torch.int32, device=dev)
            wmn, wmx = torch.full((shp[0],), float("inf"), device=dev), torch.full((shp[0],), float("-inf"), device=dev)
            tag = "x".join(str(d) for d in shp)
            add(f"weight {tag}: fake_quant per-channel", timed(_hip.TIME_FAKE_QUANT_CHANNEL, lambda i: ops.fake_quant_per_channel(
                w, ws_, wz_, 0, -32, 31)), 8 * w.numel())
            add(f"weight {tag}: MinMaxObserver per-channel (+qparams)", timed(_hip.TIME_OBSERVE_CHANNELS, lambda i: ops.observe_channels(
                w, 0, ops.UPDATE_RUNNING, 0, wmn, wmx, -32, 31, True, ops.QParamSink(ws_, wz_))), 4 * w.numel())
            if shp[0] <= 3072:
                us = timed(_hip.TIME_MSEFAST_ROWS, lambda i: ops.msefast_rows(w, 0, -8, 7, True, "no", False))
                rows[f"weight {tag}: MSEFast 4-bit symmetric per-channel (one bounded-Brent search per row)"] = {
                    "avg_us": round(us, 2), "bound": "compute (row in registers, ~15 loss evaluations per row)",
                    "algorithmic_MB": round(4 * w.numel() / 1e6, 1), "rows": shp[0]}
        # per-tensor MSEFast (configs[3] activations): one asymmetric search = 300-600 loss evaluations; resident form (one
        # persistent launch, tensor in registers) against one launch per evaluation, second call (float64 arithmetic)
        from outlier_suppression_amd.quantization.observer import AvgMSEFastObserver
        for tag, shp in (("32x128x768", (32, 128, 768)), ("32x128x3072", (32, 128, 3072))):
            xm = torch.randn(*shp, device=dev)
            xm[..., 5] *= 20
            res = {}
            for resident in (1, 0):
                ops.set_tuning("mse_resident", resident)
                try:
                    ob = AvgMSEFastObserver(bit=6, symmetric=False).to(dev)
                    ob(xm, l32, 1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ob(xm, l32, 1)
                    torch.cuda.synchronize()
                    res[resident] = ((time.perf_counter() - t0) * 1e6, int(ob.last_nfev.sum().item()))
                finally:
                    ops.set_tuning("mse_resident", 1)
            rows[f"site {tag}: AvgMSEFast per-tensor 6-bit asymmetric search (masked)"] = {
                "wall_us": round(res[1][0], 1), "loss_evaluations": res[1][1], "us_per_evaluation": round(res[1][0] / max(res[1][1], 1), 2),
                "one_launch_per_evaluation_us_per_evaluation": round(res[0][0] / max(res[0][1], 1), 2),
                "bound": "per evaluation: fp64 VALU work on the resident tensor + one exchange through memory (~2 us) + the serial Brent step (~1.5 us)"}
        # the ROUNDS kernel of an observer pass (msefast_tensor_ordered_multi_kernel): one loss evaluation of K open float64 searches per
        # launch, 604 MB per round; the loss memo off so that every timed round streams every site (with it ~1 round in 4 streams a
        # two-sided site: that is the flow's gain, not the kernel's)
        if ops.reference_sum_order("mse"):
            for tag, shp, k in (("48 x [32,128,768]", (32, 128, 768), 48), ("12 x [32,128,3072]", (32, 128, 3072), 12)):
                sites = [torch.randn(*shp, device=dev) * (1 + i % 3) for i in range(k)]
                ops.set_tuning("mse_memo", 0)
                try:
                    group = []
                    for xm in sites:
                        cur = torch.stack([xm.min(), xm.max()]).to(torch.float32)
                        group.append(ops.msefast_tensor_begin(xm, cur, None, 1, 0, 63, False, "no", True, float64_input=True))
                    ctx = ops._ordered_group_prepare(group)
                    ops._ordered_group_rounds(ctx, 2)
                    best = None
                    for _ in range(3):
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        ops._ordered_group_rounds(ctx, 12)
                        e1.record()
                        torch.cuda.synchronize()
                        us = e0.elapsed_time(e1) * 1e3 / 12
                        best = us if best is None else min(best, us)
                finally:
                    ops.set_tuning("mse_memo", 1)
                nbytes = 4 * sum(xm.numel() for xm in sites)
                rows[f"MSEFast rounds kernel, {tag} open float64 searches (one loss evaluation of each per launch, reference summation order)"] = dict(
                    hbm_row(best, nbytes, cycled=True), timer="stream events around 12 back-to-back rounds, best of 3 (kernel boundaries included)",
                    note="a bare read of the same bytes takes 100-105 us (profiles/r06_stream_pattern.txt)")
                del ctx, group, sites
        # Infinity Cache: the same 96 MiB tensor over and over (x + y = 192 MiB < 256 MiB) against the buffer cycle above
        warm_y = ev_timed(lambda i: ops.fake_quant_per_tensor(xs[0], s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4))
        add_ev("fake_quant_forward, warm (same input every launch; Infinity Cache)", warm_y, 8 * n, resident=True)
        cold_y = ev_timed(lambda i: ops.fake_quant_per_tensor(xs[i % len(xs)], s, zf, 0, 63, ops.PARAM_LSQPLUS, 1e-4))
        add_ev("fake_quant_forward, cold (4 inputs cycled, 384 MiB)", cold_y, 8 * n)
        rows["same site, eager sequence (gamma_residual, layer_norm, add, fake_quant)"] = hbm_row(seq_us, 12 * n, cycled=True)
    return rows

