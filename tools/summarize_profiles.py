#!/usr/bin/env python3
"""Summaries of tools/collect_profiles.sh: <tag>_bench_kernel_stats.md, <tag>_bench_pmc.md, roofline_traffic.json.

FETCH_SIZE / WRITE_SIZE are KB per dispatch.  On gfx950 FETCH_SIZE counts a wide coalesced streaming read
(16 B per lane) at half its bytes -- 128-byte requests tallied as 64 B (MI355X_MICROARCH.md, HBM section) --
so reads are doubled; WRITE_SIZE is taken as is (checked against the fake-quant kernel, whose writes equal
its algorithmic 96 MiB).
"""
import glob
import json
import os
import sqlite3
import sys

N_ELEM = 256 * 128 * 768
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# what the dominant kernels are compiled from: bench.py refuses a roofline_traffic.json taken from other sources
KERNEL_SOURCES = ("fused_step.h", "token_select.h", "observer.hip", "osq_device.h", "fake_quant.hip", "Makefile")


def kernel_sources_sha256():
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "outlier_suppression_amd", "csrc", name), "rb").read())
    return h.hexdigest()


def find_db(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return dbs[0] if dbs else None


def kernel_table(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total_us | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx in rows[:12]:
        lines.append(f"| {name.split('(')[0][-70:]} | {calls} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | "
                     f"{mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    return "\n".join(lines) + "\n", {r[0].split("(")[0]: r[3] / 1e3 for r in rows}


def pmc_rows(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
                       "where counter_name = ? and kernel_name like '%osq::%' group by kernel_name", (counter,)).fetchall()
    return {r[0].split("(")[0]: r[1:] for r in rows}


def main(out, tag):
    stats, avg_us = kernel_table(find_db(os.path.join(out, "trace")))
    phases = ""
    ph = os.path.join(out, f"{tag}_bench_trace_phases.md")
    if os.path.exists(ph):
        txt = open(ph).read()
        tail = txt[txt.find("| run length |"):] if "| run length |" in txt else ""
        phases = ("\nThe step's kernel by how it was launched (tools/trace_phases.py on the same database; full table: "
                  f"{tag}_bench_trace_phases.md): runs of 50 = graph replays of the K = 50 steps, longer runs = eager loops (settle, warm-up, "
                  "pre-roll), the run of 1000 = the launches that carry dispatch events (`roofline.isolated_launch_us`).  `roofline.avg_launch_us` "
                  "of the bench line is the back-to-back figure:\n\n" + tail)
    open(os.path.join(out, f"{tag}_bench_kernel_stats.md"), "w").write(
        f"# {tag}: rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --settle 0 --no-cpu-baseline --no-calib --no-kernel-table\n\n" + stats + phases)
    fetch = pmc_rows(find_db(os.path.join(out, "pmc_fetch")), "FETCH_SIZE")
    write = pmc_rows(find_db(os.path.join(out, "pmc_write")), "WRITE_SIZE")
    lines = [f"# {tag} PMC counters (rocprofv3 --pmc, separate passes), bench.py [256,128,768]", "",
             "| kernel | counter | dispatches | avg KB | min KB | max KB |", "|---|---|---|---|---|---|"]
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        short = k.split("osq::")[-1]
        f, w = fetch.get(k), write.get(k)
        if f:
            lines.append(f"| {short} | FETCH_SIZE | {f[0]} | {f[1]:.1f} | {f[2]:.1f} | {f[3]:.1f} |")
        if w:
            lines.append(f"| {short} | WRITE_SIZE | {w[0]} | {w[1]:.1f} | {w[2]:.1f} | {w[3]:.1f} |")
        rd = int(2 * f[1] * 1024) if f else None
        wr = int(w[1] * 1024) if w else None
        traffic[short] = {"FETCH_SIZE_KB_avg": f[1] if f else None, "WRITE_SIZE_KB_avg": w[1] if w else None,
                          "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                          "hbm_bytes_per_launch": (rd or 0) + (wr or 0),
                          "avg_us_kernel_trace": next((v for n, v in avg_us.items() if short in n), None)}
    lines += ["", "FETCH_SIZE is reported at half the bytes for wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section): "
              "hbm_read_bytes_per_launch in roofline_traffic.json = 2 x FETCH_SIZE.  Algorithmic bytes: the fused step "
              f"observe_fq_fused_kernel 4 B x valid elements + 8 B x {N_ELEM} (its HBM traffic should be 8 B x {N_ELEM} = {8 * N_ELEM}: "
              "x read once, y written once); fake-quant "
              f"{8 * N_ELEM} (8 B x {N_ELEM}); token_minmax 4 B x valid elements; token_select 8 B x token slots."]
    open(os.path.join(out, f"{tag}_bench_pmc.md"), "w").write("\n".join(lines) + "\n")
    traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 50 "
                        "--warmup 10 --no-cpu-baseline --no-calib`; KB per dispatch; reads doubled per MI355X_MICROARCH.md")
    traffic["_kernel_sources_sha256"] = kernel_sources_sha256()
    traffic["_profile_tag"] = tag
    json.dump(traffic, open(os.path.join(out, "roofline_traffic.json"), "w"), indent=1)
    print(stats)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
