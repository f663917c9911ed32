#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max, like --stats.
usage: python tools/rocpd_stats.py <results.db> [steps]   -> markdown table on stdout"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cur = db.cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | % | avg us | min us | max us |")
    print("|---|---|---|---|---|---|---|")
    for name, n, t, avg, mn, mx in rows:
        short = re.sub(r"\s+", " ", name.replace("lmi::", ""))[:110]
        print(f"| `{short}` | {n} | {t / 1e6:.2f} | {100 * t / tot:.1f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} |")


if __name__ == "__main__":
    main()
