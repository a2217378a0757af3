#!/usr/bin/env python3
"""Summary of one calibration flow profiled by tools/collect_calibration_profiles.sh:
<tag>_calibration_config<N>_kernel_stats.md = rocprofv3 --kernel-trace per-kernel table (top 25), the osq:: share of
the kernel time, and for the top osq:: kernels the HBM bytes per dispatch from the PMC passes (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, + WRITE_SIZE; separate passes) with the rate they imply.
usage: summarize_calibration_profile.py <out dir> <tag> <config> "<command>" """
import glob
import os
import sqlite3
import sys


def find_db(d):
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return dbs[0] if dbs else None


def main(out, tag, config, command):
    con = sqlite3.connect(find_db(os.path.join(out, "trace")))
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    osq = [r for r in rows if "osq::" in r[0]]
    osq_total = sum(r[2] for r in osq)
    gemm = sum(r[2] for r in rows if r[0].startswith("Cijk_") or "SGRO0_" in r[0] or "gemm" in r[0].lower())
    lines = [f"# {tag} calibration configs[{config}]: rocprofv3 --kernel-trace --stats -- {command}", "",
             f"kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches; **osq:: kernels {osq_total / 1e6:.1f} ms = "
             f"{100 * osq_total / total:.1f} %** ({sum(r[1] for r in osq)} dispatches); rocBLAS / hipBLASLt GEMMs {100 * gemm / total:.1f} %", "",
             "| kernel | calls | total_us | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx in rows[:25]:
        lines.append(f"| {name.split('(')[0][-80:]} | {calls} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")

    def pmc(sub, counter):
        db = find_db(os.path.join(out, sub))
        if not db:
            return {}
        c = sqlite3.connect(db)
        try:
            rs = c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? and "
                           "kernel_name like '%osq::%' group by kernel_name", (counter,)).fetchall()
        except sqlite3.Error:
            return {}
        return {r[0].split("(")[0]: (r[1], r[2]) for r in rs}
    fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
    lines += ["", "## HBM traffic of the osq:: kernels with the largest share (PMC, per dispatch; reads = 2 x FETCH_SIZE)", "",
              "| kernel | share of kernel time | avg_us | HBM read MB | HBM written MB | bytes / time |", "|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx in osq[:5]:
        key = name.split("(")[0]
        f = next((v for k, v in fetch.items() if k == key), None)
        w = next((v for k, v in write.items() if k == key), None)
        rd = 2 * f[1] * 1024 if f else None
        wr = w[1] * 1024 if w else None
        rate = f"{((rd or 0) + (wr or 0)) / (avg / 1e9) / 1e12:.2f} TB/s" if (rd is not None or wr is not None) else "n/a"
        lines.append(f"| {key[-80:]} | {100 * tot / total:.1f} % | {avg / 1e3:.2f} | {rd / 1e6:.2f} | {wr / 1e6:.2f} | {rate} |"
                     if rd is not None and wr is not None else f"| {key[-80:]} | {100 * tot / total:.1f} % | {avg / 1e3:.2f} | n/a | n/a | n/a |")
    lines += ["", "The fake-quant / observer kernels of a calibration run on SITE tensors (3-57 M elements per call, mostly 3.1 M): a launch moves",
              "8 B per element (fake-quant), 4 B per valid element (observers), 12 B (LSQ+ backward); at 4-15 us per launch the dispatch floor",
              "(4 us on this clock) is a third to a half of it -- bench.py's `kernels` table prices the same kernels at the BASELINE tensor."]
    path = os.path.join(out, f"{tag}_calibration_config{config}_kernel_stats.md")
    open(path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main(*sys.argv[1:5])
