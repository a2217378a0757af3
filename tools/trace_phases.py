#!/usr/bin/env python3
"""What a rocprofv3 --kernel-trace of bench.py says about ONE kernel, phase by phase.

A bench run issues the step's kernel in different ways -- eager loops, hipGraph replays, the event-carrying launches of
the `roofline` probe -- and the trace's one average mixes them.  This reads the rocpd database, takes the dispatches of
the kernels whose name contains <substring> in start order, splits them into runs of back-to-back launches (a new run
starts where the GPU idled longer than --idle us) and prints per run: launches, average duration (end - start of the
dispatch, what --stats averages), average period (start to next start) and average gap (end to next start).
    tools/trace_phases.py <results.db> <kernel-substring> [--idle 30] [--min-run 20]
"""
import sqlite3
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = {sys.argv[i][2:]: float(sys.argv[i + 1]) for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
    idle_ns, min_run = opt.get("idle", 30.0) * 1e3, int(opt.get("min-run", 20))
    con = sqlite3.connect(args[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_timestamp"
    e_col = "end" if "end" in cols else "end_timestamp"
    rows = con.execute(f'select "{s_col}", "{e_col}" from kernels where name like ? order by "{s_col}"', ("%" + args[1] + "%",)).fetchall()
    if not rows:
        print("no dispatches match", args[1], "columns:", cols)
        return
    runs, cur = [], [rows[0]]
    for prev, nxt in zip(rows, rows[1:]):
        if nxt[0] - prev[1] > idle_ns:
            runs.append(cur)
            cur = []
        cur.append(nxt)
    runs.append(cur)
    print(f"# {args[1]}: {len(rows)} dispatches, {len(runs)} back-to-back runs (idle > {idle_ns / 1e3:.0f} us splits); runs of >= {min_run} launches:")
    print("| run | launches | avg duration us | min | max | avg period us | avg gap us | first 3 durations us |")
    print("|---|---|---|---|---|---|---|---|")
    allD, longD = [], []
    for i, r in enumerate(runs):
        d = [(e - s) / 1e3 for s, e in r]
        allD += d
        if len(r) < min_run:
            continue
        longD += d
        per = [(b[0] - a[0]) / 1e3 for a, b in zip(r, r[1:])]
        gap = [(b[0] - a[1]) / 1e3 for a, b in zip(r, r[1:])]
        print(f"| {i} | {len(r)} | {sum(d) / len(d):.2f} | {min(d):.2f} | {max(d):.2f} | {sum(per) / len(per):.2f} | {sum(gap) / len(gap):.2f} | "
              + ", ".join(f"{v:.1f}" for v in d[:3]) + " |")
    print(f"\nall dispatches: avg {sum(allD) / len(allD):.2f} us; dispatches in the listed runs: avg {sum(longD) / max(len(longD), 1):.2f} us over {len(longD)}")
    # runs grouped by length: the bench issues K-step graph replays (runs of K or multiples) and long eager loops
    by_len = {}
    for r in runs:
        by_len.setdefault(len(r), []).extend((e - s) / 1e3 for s, e in r)
    print("\n| run length | runs | avg duration us |\n|---|---|---|")
    for n in sorted(by_len):
        cnt = len(by_len[n]) // n
        if cnt * n >= min_run:
            print(f"| {n} | {cnt} | {sum(by_len[n]) / len(by_len[n]):.2f} |")


if __name__ == "__main__":
    main()
