#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / min / max (us) and share.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]"""
import re
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    namecol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = f"select s.{namecol}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) " \
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc"
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
    for name, n, tot, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)
        lines.append(f"\"{short}\",{n},{tot / 1e3:.1f},{tot / 1e3 / n:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * tot / total:.2f}")
    lines.append(f"\"TOTAL\",{sum(r[1] for r in rows)},{total / 1e3:.1f},,,,100.0")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
