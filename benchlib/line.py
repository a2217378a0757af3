"""The ONE JSON line bench.py prints last, kept small enough for the driver to recover it from a stdout tail.

Round 5's line carried the per-kernel table and seven calibration blocks (20 KB) and the driver's 8 KB tail cut it: the
round had no graded number.  Here the line is built from the full result by `compact_line` -- headline, `roofline`,
`cpu_baseline`, a `{config: wall_s, collective_s}` calibration summary -- and bounded by `MAX_LINE_BYTES`; everything else
(`kernels`, calibration phases, `quantized_forward`, probe regions) goes to `bench_detail.json` beside bench.py and to
EARLIER stdout lines prefixed `detail ` (so that "lines starting with {" is exactly the one line).  Pure functions: no GPU,
no torch -- tests/test_bench_line.py builds a line from a stub and checks size and keys on the CPU.
"""
import json

MAX_LINE_BYTES = 4096

REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")

# config: what names the workload and says how the timed regions were issued (numbers, short strings)
_CONFIG_KEEP = ("workload", "launches_per_step", "buffers_cycled", "algorithmic_bytes_per_step", "hbm_bytes_per_step",
                "valid_token_fraction", "launch_picked", "graph_us_per_step", "eager_us_per_step", "timed_regions",
                "ms_per_step_min", "ms_per_step_max", "ms_per_step_regions", "pct_hbm_peak", "three_launch_path_ms_per_step")
_ROOFLINE_KEEP = ROOFLINE_KEYS + ("kernel", "avg_launch_us", "isolated_launch_us", "launches_timed", "duration_source",
                                  "algorithmic_bytes_per_launch", "frac_physical", "copy_rate", "frac_physical_of_copy_rate",
                                  "rocprof_avg_launch_us")
_CPU_KEEP = CPU_KEYS + ("host_cores", "full_tensor_GiB_per_s")
_COLLECTIVE_KEEP = ("backend", "ranks_seen", "world_size", "distinct_devices", "shared_gpu", "exchange_check")


def _clip(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d}


def calibration_summary(full):
    """{section: {wall_s, collective_s}} of every calibration section present in the full result."""
    out = {}
    for k, v in full.items():
        if k.startswith("calibration") and k != "calibration_summary" and isinstance(v, dict):
            if "wall_s" in v:
                out[k] = {"wall_s": v.get("wall_s"), "collective_s": v.get("collective_s")}
            elif "error" in v:
                out[k] = {"error": _clip(v["error"], 80)}
    return out


def compact_line(full):
    """The compact object of the last stdout line, from the full result dictionary."""
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    coll = full.get("collective") or {}
    line["collective"] = _pick(coll, _COLLECTIVE_KEEP)
    if isinstance(line["collective"].get("backend"), str):
        line["collective"]["backend"] = _clip(line["collective"]["backend"], 48)
    cfg = _pick(full.get("config") or {}, _CONFIG_KEEP)
    if "workload" in cfg:
        cfg["workload"] = _clip(cfg["workload"], 200)
    line["config"] = cfg
    roof = _pick(full.get("roofline") or {}, _ROOFLINE_KEEP)
    for k in ("kernel", "duration_source"):
        if k in roof:
            roof[k] = _clip(roof[k], 120)
    line["roofline"] = roof
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        cpu = _pick(cpu, _CPU_KEEP + ("error",))
        if "sample" in cpu:
            cpu["sample"] = _clip(cpu["sample"], 120)
        if "error" in cpu:
            cpu["error"] = _clip(cpu["error"], 120)
    line["cpu_baseline"] = cpu
    summary = calibration_summary(full)
    if summary:
        line["calibration_summary"] = summary
    line["detail"] = full.get("detail_file", "bench_detail.json")
    return line


def dumps_line(full):
    """JSON text of the compact line; trims optional parts, never the required ones, until it fits MAX_LINE_BYTES."""
    line = compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    for victim in (("config", "ms_per_step_regions"), ("collective", "exchange_check"), ("calibration_summary",)):
        if len(text) <= MAX_LINE_BYTES:
            break
        node = line
        for k in victim[:-1]:
            node = node.get(k) or {}
        node.pop(victim[-1], None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def check_line(text):
    """What the driver needs of the line: one JSON object, under the cap, the contract's keys, roofline and cpu_baseline
    shaped as the contract says.  Raises AssertionError naming the first thing that is wrong."""
    assert "\n" not in text, "the line must be one line"
    assert len(text) <= MAX_LINE_BYTES, f"line is {len(text)} bytes (cap {MAX_LINE_BYTES})"
    obj = json.loads(text)
    for k in REQUIRED_KEYS:
        assert k in obj, f"missing key {k!r}"
    assert isinstance(obj["config"], dict) and "workload" in obj["config"], "config.workload names the workload"
    for k in ROOFLINE_KEYS:
        assert k in obj["roofline"], f"roofline lacks {k!r}"
    if obj["cpu_baseline"] is not None and "error" not in obj["cpu_baseline"]:
        for k in CPU_KEYS:
            assert k in obj["cpu_baseline"], f"cpu_baseline lacks {k!r}"
        assert len(obj["cpu_baseline"]["sample"]) <= 120
    return obj


def detail_lines(full):
    """The sections that do not ride on the final line, one `detail <section> <json>` stdout line each."""
    skip = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "detail_file"}
    return [f"detail {k} " + json.dumps(v, separators=(",", ":")) for k, v in full.items() if k not in skip]


def parse_stdout(stdout):
    """(line object, {section: object}) from a bench.py stdout: what tests and tools read."""
    line, detail = None, {}
    for raw in stdout.splitlines():
        if raw.startswith("{"):
            line = json.loads(raw)
        elif raw.startswith("detail "):
            _, name, body = raw.split(" ", 2)
            detail[name] = json.loads(body)
    return line, detail
