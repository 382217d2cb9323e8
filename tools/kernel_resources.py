#!/usr/bin/env python
"""Register / scratch / LDS table of every kernel of libcoot_hip.so, from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed).
    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
Scratch > 0 means spills: their traffic shares vmcnt with the weight prefetch of the GEMM passes."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "coot-videotext_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result -Rpass-analysis=kernel-resource-usage".split()


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        clean = []
        for o in out[:len(names)]:
            o = o.replace("coot::", "").replace("(anonymous namespace)::", "")
            o = re.sub(r"^void ", "", o)
            m = re.match(r"([A-Za-z0-9_]+(<[^(]*>)?)", o)
            clean.append(m.group(1) if m else o)
        return clean
    except Exception:
        return names


def main():
    rows = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-device-only", "-c", f, "-o", "/dev/null"], cwd=CSRC, capture_output=True, text=True)
        cur = None
        for line in r.stderr.split("\n"):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = dict(file=f, name=m.group(1)); rows.append(cur); continue
            for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
    names = demangle([r["name"] for r in rows])
    print(f"{'file':16s} {'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>15s} {'LDS B/block':>12s} {'occupancy':>10s}")
    spilled = 0
    for r, n in zip(rows, names):
        spilled += r.get("scratch", 0) > 0
        print(f"{r['file']:16s} {n[:58]:58s} {r.get('vgpr', -1):5d} {r.get('agpr', -1):5d} {r.get('sgpr', -1):5d} {r.get('scratch', -1):15d} {r.get('lds', -1):12d} {r.get('occ', -1):10d}")
    print(f"\n{len(rows)} kernels, {spilled} with scratch")


if __name__ == "__main__":
    sys.exit(main())
