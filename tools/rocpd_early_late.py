#!/usr/bin/env python
"""Per-kernel launch duration as a function of the STEP it belongs to, from a rocprofv3 rocpd (.db) kernel trace of bench.py:
which kernels are slower in the first steps after the device was idle (round 5, verdict item 1: the driver's 20 / 5 regime).
Steps are delimited by sample_idx_kernel (one per step).  Prints, per kernel (sorted by time): average duration over steps
[2, 5], [6, 10], [11, 15], [last 5], and the ratio first / last.
Usage: python tools/rocpd_early_late.py x_results.db [out.txt]"""
import re
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    namecol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    rows = list(cur.execute(f"select s.{namecol}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    marks = [r[1] for r in rows if "sample_idx_kernel" in r[0]]
    if len(marks) < 12:
        raise SystemExit("need at least 12 steps in the trace")
    import bisect
    nsteps = len(marks)
    per = {}
    for name, s, e in rows:
        st = bisect.bisect_right(marks, s)  # step index: 0 = before the first loss section
        short = re.sub(r"\(.*", "", name).replace("coot::", "").replace("(anonymous namespace)::", "")
        short = re.sub(r"^void ", "", short)
        per.setdefault(short, {}).setdefault(st, []).append((e - s) / 1e3)
    bands = [("steps 2-5", range(2, 6)), ("6-10", range(6, 11)), ("11-15", range(11, 16)), ("last 5", range(nsteps - 5, nsteps))]
    lines = [f"# {nsteps} steps; per kernel: us per STEP (sum over its launches of a step), averaged over the band; step window = one loss section to the next",
             f"{'kernel':60s} " + " ".join(f"{b[0]:>10s}" for b in bands) + "   first/last"]
    tot = [0.0] * len(bands)
    table = []
    for k, d in per.items():
        vals = []
        for _, rg in bands:
            xs = [sum(d[s]) for s in rg if s in d]
            vals.append(sum(xs) / len(xs) if xs else 0.0)
        table.append((k, vals))
    table.sort(key=lambda r: -r[1][-1])
    for k, vals in table:
        if vals[-1] < 3:
            continue
        for i, v in enumerate(vals):
            tot[i] += v
        lines.append(f"{k[:60]:60s} " + " ".join(f"{v:10.1f}" for v in vals) + f"   {vals[0] / vals[-1] if vals[-1] else 0:8.3f}")
    lines.append(f"{'SUM':60s} " + " ".join(f"{v:10.1f}" for v in tot) + f"   {tot[0] / tot[-1] if tot[-1] else 0:8.3f}")
    gaps = [(marks[i + 1] - marks[i]) / 1e3 for i in range(len(marks) - 1)]
    lines.append("# step length (us, loss section to loss section): " + " ".join(f"{g:.0f}" for g in gaps))
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
