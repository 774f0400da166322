#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default rocprofv3 7.x output) into a per-kernel stats table
(the same columns as `rocprofv3 --stats` kernel_stats.csv).  Usage: tools/rocpd_stats.py results.db [> summary.md]"""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total_us | avg_us | min_us | max_us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx in rows[:top]:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {n} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
