#!/usr/bin/env python
"""Content hash of everything libcoot_hip.so is built from (csrc/*.hip, csrc/*.h, include/*.h, sorted by name): what a committed
counter profile (profiles/*_traffic.json, tools/pmc_traffic.py) is stamped with and what bench.py recomputes at run time — a profile
whose stamp differs from the sources of the run was taken on OTHER kernels and is reported as stale instead of being quoted
(roofline.traffic must be able to notice a traffic regression).  A commit hash cannot do this job: the profile is
committed AFTER the commit it measured."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sources_sha16(root: str = ROOT) -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "coot-videotext_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "coot-videotext_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(sources_sha16() + "\n")
