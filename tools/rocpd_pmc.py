#!/usr/bin/env python
"""Per-kernel averages of one PMC counter from a rocprofv3 rocpd (.db) counter-collection run.
Usage: python tools/rocpd_pmc.py x_results.db [out.csv]
Prints kernel, dispatches, counter name, average and total value per dispatch (FETCH_SIZE / WRITE_SIZE are in KiB)."""
import re
import sqlite3
import sys


def load(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    kcols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    evcol = "event_id" if "event_id" in kcols else "id"
    q = (f"select s.{namecol}, i.name, count(*), sum(p.value), min(p.value), max(p.value) from {pe} p "
         f"join {kd} d on p.event_id = d.{evcol} join {ks} s on d.kernel_id = s.id join {pi} i on p.pmc_id = i.id "
         f"group by s.{namecol}, i.name order by 4 desc")
    return list(cur.execute(q))


def main(db, out=None):
    rows = load(db)
    lines = ["kernel,counter,dispatches,avg_per_dispatch,total,min,max"]
    for name, cname, n, tot, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)
        lines.append(f"\"{short}\",{cname},{n},{tot / n:.2f},{tot:.1f},{mn:.2f},{mx:.2f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
