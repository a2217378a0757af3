#!/usr/bin/env python3
"""Hot-path benchmark: observer + fake-quant over BERT-base [256,128,768] activations on MI355X.

One "step" = one call of an activation quantizer on one resident fp32 batch with observer and
fake-quant both on -- AvgPruneMinMaxObserver (token-wise clipping, p = 0.95, padded tokens
skipped) -> running average -> calculate_qparams -> LSQ+ fake-quant forward -- i.e. the
``observer -> fake-quant in one call`` row of BASELINE.md.  It is ONE HIP launch
(csrc/fused_step.h: a persistent grid that keeps the tensor in registers / LDS between the
reduction and the quantisation, so x crosses HBM once).  ``value`` counts ALGORITHMIC bytes
("report against 12"): 4 B per observed (non-padded) element + 8 B per
element for the fake-quant; the HBM traffic the launch really causes is reported as
``roofline.traffic`` and is smaller (8 B per element).

Contract (driver): python bench.py --gpus N --steps K --warmup W.  N > 1: either launched by the driver as
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...``
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or called plainly as ``python bench.py --gpus N ...``, in
which case it re-launches itself under torch.distributed.run (one rank per GPU, RCCL).

Output (rank 0): the LAST stdout line is ONE compact JSON object (< 4 KB: headline, `roofline`, `cpu_baseline`, a
`{config: wall_s, collective_s}` calibration summary; benchlib/line.py).  Everything else -- the per-kernel table
(benchlib/kernels.py), the calibration flows with their phases (benchlib/calibration_flows.py), the probe regions -- is
printed BEFORE it as `detail <section> <json>` lines and written to bench_detail.json beside this file.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import SHAPE, HBM_PEAK_GBS, COPY_RATE_GBS, GIB, SHORT, make_inputs, make_quantizer, _ops_order, device_identity  # noqa: E402
from benchlib import line as bench_line  # noqa: E402


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: re-run this very command line under torch.distributed.run, one rank
    per GPU of this node (rendezvous on 127.0.0.1, a free port), and hand back its exit status.  The ranks' stdout is
    this process's stdout: rank 0 prints the one JSON line."""
    import socket
    import subprocess
    share = os.environ.get("OSQ_BENCH_SHARE_GPU") in ("1", "check")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not share:
        print(f"bench.py: --gpus {n} needs {n} visible HIP devices, this node shows {have} "
              "(OSQ_BENCH_SHARE_GPU=1 runs the N-rank control flow on one GPU with a gloo group: a test hook, not a measurement)",
              file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=float, default=1.0, help="seconds of untimed steps before the warm-up steps (0 for profiler runs)")
    ap.add_argument("--preroll", type=float, default=0.25, help="seconds of untimed graph replays directly before the timed one")
    ap.add_argument("--buffers", type=int, default=4, help="distinct input tensors cycled through (4 x 96 MiB > 256 MiB Infinity Cache)")
    ap.add_argument("--eager", action="store_true", help="time the eager loop of module calls instead of the captured graph (= --launch eager)")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager"],
                    help="how the K timed module calls are issued: one hipGraph replay, the eager loop, or (auto) whichever of the two "
                         "measures faster in untimed K-step regions directly before the timed one")
    ap.add_argument("--regions", type=int, default=5, help="consecutive timed regions of exactly K steps each; ms_per_step is their median (min / max reported beside it)")
    ap.add_argument("--kernel-launches", type=int, default=1000, help="launches of the step's kernel timed by HIP events on their dispatch packets for `roofline` (a sustained back-to-back run, the regime rocprofv3 --kernel-trace averages over)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-calib", action="store_true", help="skip the 256-sample calibration wall-clock section")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the per-kernel timing table")
    ap.add_argument("--calib-search", default="cached", choices=["cached", "literal"])
    ap.add_argument("--calib-configs", default="0,1,2,3,4,5", help="BASELINE configs whose calibration wall-clock is measured (4 = configs[4] at the bart-LARGE dimensions BASELINE.json names, ~1 min; 5 = the same flow at bart-base dimensions, what the reference's shipped config uses)")
    args = ap.parse_args()

    import torch.distributed as dist
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    # test hook (1-GPU box): OSQ_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with a gloo group so that the
    # N > 1 control flow (sharding, exchange, barriers, max-over-ranks clock) can be exercised without N GPUs
    # ("check": the same hook, but the one-process-per-GPU assertion below stays armed -- tests/test_gpu_sharded.py uses it to
    # see the command refuse N ranks on one device)
    share = os.environ.get("OSQ_BENCH_SHARE_GPU") in ("1", "check")
    enforce_devices = not share or os.environ.get("OSQ_BENCH_SHARE_GPU") == "check"
    if share:
        local_rank = 0
        # several processes on ONE GPU: the one-launch step wants every CU for itself and cannot be ordered against
        # another process's launch -- the shared-GPU hook times the three-launch path
        os.environ["OSQ_FUSED_STEP"] = "0"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)    # "nccl" is RCCL on ROCm

    collective = {"backend": None, "ranks_seen": 1}
    if world > 1:
        # every rank contributes its rank number through the backend the data path uses: the gathered tensor proves that
        # N distinct ranks took part in an RCCL (backend "nccl") collective -- or says that the gloo test hook ran instead
        probe = torch.tensor([rank], dtype=torch.int64, device="cpu" if share else dev)
        seen = torch.empty(world, dtype=torch.int64, device=probe.device)
        dist.all_gather_into_tensor(seen, probe)
        # ... and every rank's DEVICE: N ranks must sit on N different GPUs (a mis-set LOCAL_RANK / visibility mask would put
        # two on one, where RCCL fails late or not at all), and the data path's backend must be RCCL -- fail here, loudly
        idents = [None] * world
        dist.all_gather_object(idents, device_identity(dev))
        distinct = len(set(idents))
        collective = {"backend": "rccl (torch.distributed backend 'nccl')" if dist.get_backend() == "nccl" else dist.get_backend() + " (shared-GPU test hook)",
                      "ranks_seen": int(seen.unique().numel()), "world_size": dist.get_world_size(),
                      "devices_visible": torch.cuda.device_count(), "distinct_devices": distinct, "shared_gpu": share}
        if int(seen.unique().numel()) != world:
            raise SystemExit(f"bench.py: {world} ranks were launched but only {int(seen.unique().numel())} took part in the first collective")
        if not share and dist.get_backend() != "nccl":
            raise SystemExit(f"bench.py: --gpus {world} must run on RCCL (torch.distributed backend 'nccl'), got {dist.get_backend()!r}")
        if enforce_devices:
            if distinct != world:
                raise SystemExit(f"bench.py: {world} ranks share {distinct} device(s): {idents} -- one process per GPU is the contract "
                                 "(check LOCAL_RANK / HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES)")
    from outlier_suppression_amd import _hip, calibration
    _hip.load()
    for kv in filter(None, os.environ.get("OSQ_BENCH_TUNING", "").split(",")):      # A/B runs: "key=value,key=value" through osq_set_tuning
        from outlier_suppression_amd import ops as _ops
        _ops.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    q = make_quantizer(dev)
    xs, lengths = make_inputs(dev, args.buffers, seed=1234 + rank)
    lengths = lengths.to(dev)
    n_elem = xs[0].numel()
    valid_elem = int(lengths.sum().item()) * SHAPE[2]
    bytes_step = 4 * valid_elem + 8 * n_elem

    # N > 1: every step also leaves ITS batch's (min, max) in row i of the table the ranks exchange after the timed steps
    # (the one-launch step writes it on the way: osq_observe_tokens_fake_quant, cur_minmax)
    table = torch.zeros(args.steps, 1, 2, device=dev)
    rows = [table[i, 0] for i in range(args.steps)]
    record = world > 1

    def step(i):
        if record:
            q.observer.__dict__["_record"] = rows[i % args.steps]
        return q(xs[i % len(xs)], lengths, 1)

    with torch.no_grad():
        # settle (untimed, before the W warm-up steps): on a freshly started box the first seconds of any Python process
        # are slowed by the image still paging in, and the GPU clocks ramp.  One launch per step: ~15 us of host work
        # against ~47 us of GPU work, so the timed loop is GPU-bound from its second step on
        # The untimed loops keep the result in `y` exactly like the timed loop does: the previous output is still
        # alive while the next one is allocated, so the caching allocator needs TWO 96 MiB blocks.  (Round 1 dropped
        # the result here and kept it in the timed loop: the second block was hipMalloc'ed inside the timed region,
        # 22 % of a 20-step run.)
        t_settle = time.perf_counter()
        i = 0
        y = None
        while time.perf_counter() - t_settle < args.settle:
            y = step(i)
            i += 1
        torch.cuda.synchronize()
        for i in range(args.warmup):
            y = step(i)
    if world > 1:   # untimed: the first collective of a process group builds the RCCL communicator
        calibration.gather_batch_table(table, args.steps * world)
    import ctypes
    lib = _hip.load()

    def barrier(done=None):
        if done is not None:                   # poll the stream's last event first: a blocking synchronize wakes the host
            while not done.query():            # ~30 us after the GPU has finished (1.5 us per step of a 20-step region)
                pass
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the K timed steps are K module calls, issued either as ONE replay of a hipGraph captured after the warm-up (the
    # GPU executes exactly the launches the eager loop issues: same kernels, same arguments, same order) or as the eager
    # loop itself.  Which of the two gets a 0.8 ms region through faster is a property of the lease: round 3's driver
    # box read 44.1 us per step for the replay and 39.5 for the eager loop in the same run, the builder's boxes 40.0 and
    # 51.5 (an eager region that starts on an idle GPU).  So both are prepared, both are measured in UNTIMED K-step
    # regions directly before the timed one, and the timed region -- one region, exactly K steps, bracketed as the
    # contract says -- is issued the way that measured faster here (--launch graph / eager forces one).
    if args.eager:
        args.launch = "eager"
    graph, graph_ws, graph_note = None, None, ""
    if args.launch != "eager":
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for i in range(3):                           # allocator + workspace of the capture stream
                    y = step(i)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph, stream=side):
                graph_ws = _hip.workspace(dev)
                for i in range(args.steps):
                    y = step(i)
            torch.cuda.synchronize()
            graph.replay()                                   # untimed: the first replay of a graph also uploads it (+4 us/step at K = 20)
            torch.cuda.synchronize()
        except Exception as e:                               # capture unavailable: time the eager loop
            graph, graph_note = None, f" (graph capture failed: {type(e).__name__})"

    import gc
    gc.collect()
    gc.disable()                               # no collector pauses inside the timed region

    def issue(mode):
        if mode == "graph":
            graph.replay()
            return None
        out = None
        for i in range(args.steps):
            out = step(i)                      # the module call itself, nothing else in the loop
        return out

    def region(mode, exchange=False):
        """One K-step region: barrier + synchronize, K module calls, (N > 1: the exchange), synchronize + barrier.  Seconds."""
        before = torch.cuda.Event()            # poll, do not block: a host thread that slept in hipStreamSynchronize for the
        before.record()                        # 60 ms of the pre-roll wakes up cold and issues the region ~30 us late
        barrier(before)                        # (measured: +1.5 us per step at K = 20 against regions that follow a short wait)
        t0 = time.perf_counter()
        with torch.no_grad():                  # the reference calibrates under no_grad (token_wise_clipping.py:29-47)
            out = issue(mode)
        host = time.perf_counter() - t0
        gathered = None
        if exchange and world > 1:
            # the path's one real exchange: the per-batch rows the K steps have just recorded, gathered once
            gathered = calibration.gather_batch_table(table, args.steps * world)
        done = torch.cuda.Event()
        done.record()
        barrier(done)
        return time.perf_counter() - t0, host, gathered, out

    if world > 1:                              # the ranks issue the same way: a rank whose capture failed takes every rank to the eager loop
        ok = torch.tensor([1 if graph is not None else 0], dtype=torch.int64, device="cpu" if share else dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            graph = None
    modes = ["graph", "eager"] if (graph is not None and args.launch == "auto") else (["graph"] if graph is not None else ["eager"])
    # untimed: capture and the collector pass above leave the GPU idle for tens of milliseconds and its clocks fall back
    # (MI355X_MICROARCH.md, DVFS; tools/region_probe.py: the same 20-step replay reads 41.5 us per step in the first
    # milliseconds and 40.3 after ~50 ms of this work).  The timed region should measure the device in the state it
    # runs this work in: issue the steps for --preroll seconds, in every mode that may be timed
    with torch.no_grad():
        for mode in modes:
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < args.preroll / len(modes):
                for _ in range(4):
                    y = issue(mode)
                torch.cuda.synchronize()
    probes = {m: [] for m in modes}
    if len(modes) > 1:
        for _ in range(5):                     # alternating, so that neither mode owns the warmer half
            for m in modes:
                probes[m].append(region(m)[0])
        med = {m: sorted(v)[len(v) // 2] for m, v in probes.items()}
        pick = min(modes, key=lambda m: med[m])
        if world > 1:                          # one decision for the job: rank 0's
            flag = torch.tensor([modes.index(pick)], dtype=torch.int64, device="cpu" if share else dev)
            dist.broadcast(flag, src=0)
            pick = modes[int(flag.item())]
    else:
        pick = modes[0]
    with torch.no_grad():
        # untimed: ~12 ms of the chosen mode back to back (no synchronisation in between) directly in front of the timed
        # region.  Measured (tools/region_probe2.py, 20-step regions, us per step): after an idle gap of 5 ms 40.9, after 3-20
        # replays 39.2-39.8, after 75 replays (60 ms) 39.8-40.9, after 300 replays 40.6-41.0 -- the clocks sag when the GPU
        # idles AND when it has been busy for tens of milliseconds; a short run-up is the state the path's launches meet
        for _ in range(max(3, int(0.012 / (args.steps * 40e-6)))):
            y = issue(pick)
    # ---- the timed regions: `--regions` CONSECUTIVE regions (default 5) of exactly K steps each, every one bracketed by barrier +
    # synchronize on both sides and (N > 1) closed by the path's exchange; per region the MAX over ranks; `ms_per_step` and
    # `value` are the MEDIAN region (min / max and every region's figure ride beside it: round 5 reported one region)
    region_s, gathered = [], None
    for _ in range(max(1, args.regions)):
        dt_r, host_dt, gathered, y = region(pick, exchange=True)
        region_s.append(dt_r)
    gc.enable()
    if world > 1:
        t = torch.tensor(region_s, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = t.tolist()
    dt = sorted(region_s)[len(region_s) // 2]
    probe_stats = {m: {"median": round(sorted(v)[len(v) // 2] / args.steps * 1e6, 3), "min": round(min(v) / args.steps * 1e6, 3),
                       "max": round(max(v) / args.steps * 1e6, 3), "regions": len(v)} for m, v in probes.items() if v}
    launch_mode = ("hipGraph replay of %d captured module calls" % args.steps if pick == "graph" else "eager loop of %d module calls" % args.steps) + graph_note
    if len(modes) > 1:
        launch_mode += "; chosen by untimed probes of both (median of 5 regions, us per step: " + \
                       ", ".join(f"{m} {med[m] / args.steps * 1e6:.2f}" for m in modes) + ")"
    exchange_check = None
    if world > 1:
        # What the exchange is FOR: every rank replays the gathered rows in global batch order and must end up with the
        # statistic a single process computes from the same table (observer.py:194-202: m <- (m * cnt + cur) / (cnt + 1),
        # fp32, sequential).  Checked here on every rank: (a) the device replay (osq_replay_statistics, what
        # calibrate_sharded runs) equals the host's fp32 restatement of that loop on the gathered table bit for bit,
        # (b) all ranks hold the same bits, (c) the rows a rank contributed are the rows it recorded.
        import numpy as np
        rq = make_quantizer(dev)
        calibration.replay(gathered, [("bench.act_fake_quant", rq)], fresh=True)
        torch.cuda.synchronize()
        g_host = gathered.detach().cpu().numpy().astype(np.float32)
        mn = mx = None
        for j in range(g_host.shape[0]):
            cmin, cmax = g_host[j, 0]
            if mn is None:
                mn, mx = cmin, cmax
            else:
                c = np.float32(j)
                mn = np.float32(np.float32(np.float32(mn * c) + cmin) / np.float32(j + 1))
                mx = np.float32(np.float32(np.float32(mx * c) + cmax) / np.float32(j + 1))
        got = (np.float32(rq.observer.min_val.item()), np.float32(rq.observer.max_val.item()))
        mine = table.detach().cpu().numpy()
        own_rows_ok = bool(np.array_equal(g_host[rank::world][:args.steps], mine)) and bool(np.isfinite(mine).all()) and bool((mine[:, 0, 0] < mine[:, 0, 1]).all())
        vec = torch.tensor([float(got[0]), float(got[1]), float(rq.scale.item()), float(rq.zero_point.item())], dtype=torch.float64,
                           device="cpu" if share else dev)
        every = torch.empty(world * 4, dtype=torch.float64, device=vec.device)
        dist.all_gather_into_tensor(every, vec)
        same_on_all = bool((every.view(world, 4) == vec.view(1, 4)).all().item())
        replay_ok = bool(got[0] == mn and got[1] == mx)
        exchange_check = {"rows_gathered": int(g_host.shape[0]), "replay_equals_one_process_loop": replay_ok,
                          "same_bits_on_every_rank": same_on_all, "own_rows_intact": own_rows_ok,
                          "replayed_min_max": [float(got[0]), float(got[1])]}
        if not (replay_ok and same_on_all and own_rows_ok):
            raise SystemExit(f"bench.py: the exchanged statistics do not replay to the one-process result: {exchange_check}")

    def check_status():
        # every persistent launch of this process (timed replay, eager loop) marked its workspace: one call reads them all
        try:
            ops_mod.check_persistent("bench.py")
        except ops_mod.PersistentLaunchTimeout as e:
            raise SystemExit(f"bench.py: {e}; results are invalid")

    from outlier_suppression_amd import ops as ops_mod
    check_status()
    # the same K steps as an eager loop (untimed region): GPU time per step and host enqueue time per step
    with torch.no_grad():
        eager_runs = []
        for _ in range(5):                     # median of five K-step loops (a single 1 ms loop is at the mercy of its first call)
            torch.cuda.synchronize()
            te = time.perf_counter()
            for i in range(args.steps):
                y = step(i)
            eh = time.perf_counter() - te
            torch.cuda.synchronize()
            eager_runs.append((time.perf_counter() - te, eh))
        eager_runs.sort()
        eager_dt, eager_host = eager_runs[len(eager_runs) // 2]
    del y
    check_status()

    # ---- roofline of the step's kernel.  Two HIP-event clocks, both on the stream the kernel runs on:
    #  (1) BACK TO BACK, the regime of the timed regions (and of most of a rocprofv3 --kernel-trace of this command): stream
    #      events around R x K launches issued exactly as a timed region issues them (the picked mode), the first event recorded
    #      while the GPU is still busy with earlier launches.  Elapsed / launches is the launch PERIOD, and back to back the
    #      period IS the duration: the trace of this command (profiles/r06_bench_trace_phases.md) shows every graph-replayed or
    #      eagerly queued launch starting where its predecessor ends (gap 0.00-0.25 us), durations 37.9-38.1 us.  `roofline`
    #      is priced with THIS duration -- it reproduces from the committed trace.
    #  (2) ISOLATED: events riding on the dispatch packet of each of --kernel-launches launches (hipExtLaunchKernelGGL inside
    #      the library).  A launch that carries events is fenced off from its neighbours and meets a drained memory system: 36.9 us
    #      in the same run, the same figure the trace gives for exactly these launches.  Rounds 1-5 priced `roofline` with this
    #      one (5 % kinder than the trace average); it is reported as `isolated_launch_us`.
    reps_b2b = max(1, -(-args.kernel_launches // args.steps))
    b2b = []
    with torch.no_grad():
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2):
                issue(pick)                                  # the queue is full when the first event is recorded
            e0.record()
            for _ in range(reps_b2b):
                issue(pick)
            e1.record()
            torch.cuda.synchronize()
            b2b.append(e0.elapsed_time(e1) / (reps_b2b * args.steps))      # ms per launch
    b2b.sort()
    b2b_ms = b2b[len(b2b) // 2]
    check_status()
    n_probe = max(args.steps, args.kernel_launches)
    pairs = []
    for _ in range(n_probe):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _hip.check(lib.osq_timing_events_create(ctypes.byref(a), ctypes.byref(b)), "timing_events_create")
        pairs.append((a, b))
    fused_on = os.environ.get("OSQ_FUSED_STEP", "1") != "0"      # off: ranks sharing one GPU (test hook) run three launches
    probe_family = _hip.TIME_FUSED_STEP if fused_on else _hip.TIME_FAKE_QUANT
    with torch.no_grad():
        for i in range(n_probe):
            lib.osq_time_next_launch(probe_family, *pairs[i])
            q(xs[i % len(xs)], lengths, 1)
    torch.cuda.synchronize()
    k_seq = []
    for a, b in pairs:
        us = ctypes.c_float()
        _hip.check(lib.osq_timing_elapsed_us(a, b, ctypes.byref(us)), "timing_elapsed_us")
        k_seq.append(us.value * 1e-3)
        lib.osq_timing_events_destroy(a, b)
    k_ms = sorted(k_seq)
    iso_avg_ms = sum(k_ms) / len(k_ms)
    probe_bytes = bytes_step if fused_on else 8 * n_elem
    # one launch per step only with the one-launch step; the three-launch path (ranks sharing a GPU: test hook) is priced by its
    # fake-quant launch's dispatch events as before
    k_avg_ms = b2b_ms if fused_on else iso_avg_ms
    achieved = probe_bytes / (k_avg_ms * 1e-3) / 1e9
    # roofline.traffic: HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, separate
    # rocprofv3 passes: tools/collect_profiles.sh).  Counters cannot be read from inside this process, so the figure comes
    # from the committed profile -- but only while that profile was taken from the SAME kernel sources: the file carries the
    # hash of the sources it measured, and a mismatch reports null with the reason instead of a number that went stale
    traffic, traffic_source, rocprof_avg_us = None, None, None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from summarize_profiles import kernel_sources_sha256
            table_j = json.load(open(tpath))
            now, then = kernel_sources_sha256(), table_j.get("_kernel_sources_sha256")
            if then == now:
                entry = next((v for k, v in table_j.items()
                              if k.startswith("observe_fq_fused_kernel" if fused_on else "fq_tensor_vec_kernel") and isinstance(v, dict)), {})
                traffic = entry.get("hbm_bytes_per_launch")
                rocprof_avg_us = round(entry["avg_us_kernel_trace"], 2) if entry.get("avg_us_kernel_trace") else None
                traffic_source = (f"profiles/roofline_traffic.json ({table_j.get('_profile_tag', '?')}: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, "
                                  f"same command; kernel sources sha256 {now[:12]} = the ones built here)")
            else:
                traffic_source = (f"none: profiles/roofline_traffic.json was measured on other kernel sources (sha256 {str(then)[:12]} vs {now[:12]} here); "
                                  "re-run tools/collect_profiles.sh")
        except Exception as e:
            traffic, traffic_source = None, f"none: {type(e).__name__}: {e}"[:200]

    # the same step as three launches (token_minmax, token_select, fake_quant), for reference
    from outlier_suppression_amd import ops
    ops.set_tuning("fused_step", 0)
    q3 = make_quantizer(dev)
    with torch.no_grad():
        for i in range(20):
            q3(xs[i % len(xs)], lengths, 1)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for i in range(max(args.steps, 50)):
            q3(xs[i % len(xs)], lengths, 1)
        torch.cuda.synchronize()
        three_ms = (time.perf_counter() - t3) / max(args.steps, 50) * 1e3
    ops.set_tuning("fused_step", 1)

    value = bytes_step * args.steps * world / dt / GIB
    per_step_us = [round(r / args.steps * 1e6, 3) for r in region_s]
    out = {
        "metric": "fake-quant+observer GiB/s (% HBM peak)",
        "value": round(value, 2),
        "unit": "GiB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "collective": dict(collective, exchange_check=exchange_check),
        "config": {"workload": "configs[1] site at the metric's size: BERT-base activation [256,128,768] fp32, AvgPruneMinMaxObserver(p=0.95, "
                               "lengths randint(8,129)) -> running mean -> qparams -> LSQ+ fake-quant W6A6 asym [0,63]",
                   "launches_per_step": 1 if fused_on else 3, "buffers_cycled": len(xs),
                   "algorithmic_bytes_per_step": bytes_step, "valid_token_fraction": round(valid_elem / n_elem, 4),
                   "hbm_bytes_per_step": 8 * n_elem,
                   "launch": launch_mode, "launch_picked": pick,
                   # ms_per_step is the MEDIAN of `timed_regions` consecutive K-step regions; their spread, and every region
                   "timed_regions": len(region_s), "ms_per_step_min": round(min(region_s) / args.steps * 1e3, 5),
                   "ms_per_step_max": round(max(region_s) / args.steps * 1e3, 5), "ms_per_step_regions": [round(u * 1e-3, 5) for u in per_step_us],
                   # the untimed probe regions that decided how the timed regions are issued (five K-step regions per mode, alternating),
                   # as numbers: median / min / max microseconds per step of each mode (null: that mode was not probed)
                   "graph_us_per_step": probe_stats.get("graph", {}).get("median"), "eager_us_per_step": probe_stats.get("eager", {}).get("median"),
                   "probe_regions_us_per_step": probe_stats,
                   "sum_tier": ("package default: MSEFast losses in ATen's one-thread order, LSQ / LSQ+ parameter gradients order-free; the step "
                                "itself holds no such sum" if _ops_order() else "order-free (OSQ_STRICT=0)"),
                   "eager_ms_per_step": round(eager_dt / args.steps * 1e3, 5),
                   "host_enqueue_ms_per_step": round(eager_host / args.steps * 1e3, 5),
                   "three_launch_path_ms_per_step": round(three_ms, 5),
                   "pct_hbm_peak": round(100.0 * value * GIB / 1e9 / (HBM_PEAK_GBS * world), 2)},
        "roofline": {"bound": "hbm", "kernel": ("observe_fq_fused_kernel<3>: per-token extrema + token-wise clipping + running mean + qparams + "
                                                "fake-quant, ONE launch")
                     if fused_on else "fq_tensor_vec_kernel (fake-quant forward of the three-launch path, 8 B per elem)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "duration_source": (f"stream events around {reps_b2b * args.steps} back-to-back launches issued like the timed regions ({pick}), median of 3"
                                         if fused_on else f"dispatch events of {len(k_ms)} launches"),
                     "isolated_launch_us": round(iso_avg_ms * 1e3, 2),
                     # the committed rocprofv3 --kernel-trace average of the same kernel (same sources, by hash): must agree with avg_launch_us
                     "rocprof_avg_launch_us": rocprof_avg_us,
                     # the same launch priced by the bytes that physically cross HBM (PMC): x is read once and y written
                     # once because the tensor stays on chip between the reduction and the quantisation -- `frac` counts
                     # the ALGORITHMIC 4 B per observed + 8 B per element of SURVEY 8d, `frac_physical` the 8 B that move
                     "achieved_physical": round(traffic / (k_avg_ms * 1e-3) / 1e9, 1) if traffic else None,
                     "frac_physical": round(traffic / (k_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                     # what a float4 copy kernel reaches on this part (MI355X_MICROARCH.md, chip-level parameters: 6.29 TB/s
                     # measured = 79 % of the 8 TB/s specification): the physical rate as a fraction of THAT
                     "copy_rate": COPY_RATE_GBS,
                     "frac_physical_of_copy_rate": round(traffic / (k_avg_ms * 1e-3) / 1e9 / COPY_RATE_GBS, 4) if traffic else None,
                     "avg_launch_us": round(k_avg_ms * 1e3, 2), "back_to_back_runs_us": [round(v * 1e3, 2) for v in b2b],
                     "isolated_median_us": round(k_ms[len(k_ms) // 2] * 1e3, 2), "isolated_min_us": round(k_ms[0] * 1e3, 2),
                     "isolated_max_us": round(k_ms[-1] * 1e3, 2),
                     "launches_timed": reps_b2b * args.steps if fused_on else len(k_ms), "algorithmic_bytes_per_launch": probe_bytes},
        "detail_file": os.path.basename(args.detail_file),
    }

    def emit(final=False):
        """rank 0: the detail file (rewritten as sections arrive: a later crash keeps what was measured) and, at the end, the
        detail lines followed by THE line -- the last thing on stdout."""
        if rank != 0:
            return
        try:
            with open(args.detail_file, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            print(f"bench.py: could not write {args.detail_file}: {e}", file=sys.stderr)
        if final:
            for ln in bench_line.detail_lines(out):
                print(ln)
            text = bench_line.dumps_line(out)
            bench_line.check_line(text)
            print(text, flush=True)

    emit()
    if rank == 0 and not args.no_kernel_table:
        from benchlib.kernels import kernel_table
        try:          # a failure in the per-kernel table must not cost the headline line
            out["kernels"] = kernel_table(dev, xs, lengths)
        except Exception as e:
            out["kernels"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        emit()
    if not args.no_calib:
        from benchlib.calibration_flows import calibration_plain, calibration_wall_clock, calibration_extra, quantized_forward_times
        wanted = {int(c) for c in args.calib_configs.split(",") if c.strip()}
        if 0 in wanted:
            try:
                out["calibration_config0"] = calibration_plain(dev, rank, world)
            except Exception as e:
                if world > 1:
                    raise          # the other ranks are inside its collectives: fail the job rather than hang it
                out["calibration_config0"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if 1 in wanted:
            try:
                out["calibration"] = calibration_wall_clock(dev, rank, world, args.calib_search)
            except Exception as e:
                if world > 1:
                    raise
                out["calibration"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        emit()
        for which in (2, 3, 4, 5):
            if which not in wanted:
                continue
            try:          # a failure here must not cost the headline line
                out["calibration_config4_bart_base" if which == 5 else f"calibration_config{which}"] = calibration_extra(dev, rank, world, which)
            except Exception as e:
                if world > 1:
                    raise          # see above: never leave the other ranks waiting in a collective
                out["calibration_config4_bart_base" if which == 5 else f"calibration_config{which}"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
            emit()
        # Metric 2 in one place: wall-clock and the collective's share of every measured config at the launched N
        out["calibration_summary"] = {
            "n_gpus": world,
            "configs": {k: {"wall_s": v.get("wall_s"), "collective_s": v.get("collective_s", 0.0 if world == 1 else None),
                            "collective_phases_s": v.get("collective_phases_s"),
                            "same_parameters_on_every_rank": (v.get("exchange_check") or {}).get("same_bits_on_every_rank")}
                        for k, v in out.items() if k.startswith("calibration") and isinstance(v, dict) and "wall_s" in v}}
        if rank == 0 and not SHORT:
            try:
                out["quantized_forward"] = quantized_forward_times(dev)
            except Exception as e:
                out["quantized_forward"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from benchlib.cpu import cpu_baseline
        try:
            out["cpu_baseline"] = cpu_baseline(1234, args.cpu_budget)
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    elif rank == 0:
        out["cpu_baseline"] = None
    emit(final=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

