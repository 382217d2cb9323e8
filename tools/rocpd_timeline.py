#!/usr/bin/env python
"""Timeline of the LAST training step in a rocprofv3 rocpd (.db) kernel trace: one line per kernel dispatch with its
queue (= HIP stream), start offset and duration, plus per-queue busy time and the idle gaps on the busiest queue.
A step is delimited by sample_idx_kernel (the draw of the cycle-consistency positions: exactly one per step; else every second
cl_norm_kernel): the window shown runs from one loss section to the next, i.e. backward of step n followed by forward of step n + 1.
Usage: python tools/rocpd_timeline.py x_results.db [out.txt]"""
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
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    gcols = [c for c in ("grid_size_x", "workgroup_size_x") if c in cols]
    sel = f"s.{namecol}, d.start, d.end, " + (f"d.{qcol}" if qcol else "0") + "".join(f", d.{c}" for c in gcols)
    rows = list(cur.execute(f"select {sel} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    # exactly one per step: the draw of the cycle-consistency positions; without that loss, the contrastive normalisation (two
    # launches per step since the loss runs in two parts: every second one)
    ad = [i for i, r in enumerate(rows) if "sample_idx_kernel" in r[0]]
    if len(ad) < 3:
        ad = [i for i, r in enumerate(rows) if "cl_norm_kernel" in r[0]][::2]
    if len(ad) < 3:
        raise SystemExit("need at least 3 steps in the trace")
    lo, hi = ad[-2] + 1, ad[-1] + 1
    step = rows[lo:hi]
    t0 = rows[ad[-2]][2]
    lines = [f"# step: {len(step)} dispatches, {(step[-1][2] - t0) / 1e3:.1f} us from one loss section to the next"]
    busy = {}
    for r in step:
        name = re.sub(r"\(.*", "", r[0]).replace("coot::", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"^void ", "", name)
        q = r[3]
        busy[q] = busy.get(q, 0) + (r[2] - r[1])
        wg = f" wgs={r[4] // r[5]}" if len(r) >= 6 and r[5] else ""
        lines.append(f"{(r[1] - t0) / 1e3:9.1f} us  +{(r[2] - r[1]) / 1e3:7.1f}  q{q}{wg}  {name[:90]}")
    lines.append("# busy time per queue (us): " + ", ".join(f"q{q}: {v / 1e3:.1f}" for q, v in sorted(busy.items())))
    union = 0
    cur_s, cur_e = None, None
    for r in step:
        if cur_e is None or r[1] > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = r[1], r[2]
        else:
            cur_e = max(cur_e, r[2])
    union += cur_e - cur_s
    lines.append(f"# GPU non-idle (union over queues): {union / 1e3:.1f} us; sum of kernel durations {sum(busy.values()) / 1e3:.1f} us")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
