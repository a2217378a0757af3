#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd SQLite, ROCm 7.2 default output) into the per-kernel
summary table committed under profiles/ (same columns as `rocprofv3 --stats` CSV)."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    by_grid = "--by-grid" in sys.argv
    key = "name, grid_x, grid_y" if by_grid else "name"
    rows = con.execute(f"select name, count(*), sum(duration), avg(duration), min(duration), max(duration)"
                       f"{', grid_x, grid_y' if by_grid else ''} from kernels group by {key} "
                       f"order by {'name, grid_x' if by_grid else 'sum(duration) desc'}").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total_us | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx, *grid in rows:
        short = name.split("(")[0][-70:] + (f" grid={grid[0]}x{grid[1]}" if grid else "")
        lines.append(f"| {short} | {calls} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*[a for a in sys.argv[1:] if not a.startswith('--')][:2])
