#!/usr/bin/env python
"""MFMA utilisation per kernel from two rocprofv3 PMC passes of `python bench.py` (tools/profile_round.sh):
  pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE      pass B: SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 SQ_WAVE_CYCLES
reduced by tools/rocpd_pmc.py (one row per kernel and counter: rows, average, total over all rows).

A dispatch yields one row per shader engine for the SQ counters (32 on MI355X) and one per XCD for GRBM_GUI_ACTIVE (8), so
  MfmaUtil [%]   = sum_SE(SQ_VALU_MFMA_BUSY_CYCLES) / (mean_XCD(GRBM_GUI_ACTIVE) * 1024 SIMDs) * 100      (rocprofv3's derived metric)
  MFMA FLOPs     = sum_SE(SQ_INSTS_VALU_MFMA_MOPS_BF16) * 512
  achieved TF/s  = MFMA FLOPs / (GRBM_GUI_ACTIVE cycles / 2.4 GHz)   (counter-derived; the kernel-trace duration is in kernel_stats_*.csv)
Usage: python tools/pmc_mfma.py busy.csv mops.csv out.json"""
import csv
import json
import re
import sys

SE, XCD, SIMDS, CLK = 32, 8, 1024, 2.4e9


def read(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out.setdefault(r["kernel"], {})[r["counter"]] = (int(r["dispatches"]), float(r["total"]))
    return out


def short(k):
    m = re.search(r"(\d+)([a-z_0-9]+_kernel)(ILi(\d+)E)?", k)
    if not m:
        return k
    name = m.group(2)
    name = re.sub(r"^coot\d+", "", name)
    return name + (f"<{m.group(4)}>" if m.group(4) else "")


def main(busy_csv, mops_csv, out_json):
    A, B = read(busy_csv), read(mops_csv)
    res = {"source": [busy_csv, mops_csv], "note": __doc__.split("Usage")[0].strip(), "kernels": {}}
    for k, a in A.items():
        if "GRBM_GUI_ACTIVE" not in a or "SQ_VALU_MFMA_BUSY_CYCLES" not in a:
            continue
        nd = a["GRBM_GUI_ACTIVE"][0] // XCD
        if nd == 0:
            continue
        gui = a["GRBM_GUI_ACTIVE"][1] / a["GRBM_GUI_ACTIVE"][0]          # mean cycles the chip was busy with this dispatch
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"][1] / nd                         # summed over the shader engines, per dispatch
        e = {"dispatches": nd, "gui_active_cycles": round(gui), "mfma_busy_cycles_sum": round(busy),
             "mfma_util_pct": round(100.0 * busy / (gui * SIMDS), 2)}
        b = B.get(k, {})
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in b:
            ndb = max(b["SQ_INSTS_VALU_MFMA_MOPS_BF16"][0] // SE, 1)
            fl = b["SQ_INSTS_VALU_MFMA_MOPS_BF16"][1] / ndb * 512.0
            e["mfma_gflop_per_dispatch"] = round(fl / 1e9, 3)
            e["tflops_from_counters"] = round(fl / (gui / CLK) / 1e12, 1)
            e["frac_of_2500_tflops"] = round(fl / (gui / CLK) / 2.5e15, 4)
        if e["mfma_busy_cycles_sum"] > 0:
            res["kernels"][short(k)] = e
    res["kernels"] = dict(sorted(res["kernels"].items(), key=lambda kv: -kv[1]["gui_active_cycles"] * kv[1]["dispatches"]))
    json.dump(res, open(out_json, "w"), indent=1)
    for k, e in res["kernels"].items():
        print(f"{k:32s} x{e['dispatches']:3d}  MfmaUtil {e['mfma_util_pct']:6.2f} %  {e.get('tflops_from_counters', 0):7.1f} TF/s  {e.get('mfma_gflop_per_dispatch', 0):8.2f} GFLOP/launch")


if __name__ == "__main__":
    main(*sys.argv[1:4])
