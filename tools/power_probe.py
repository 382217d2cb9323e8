#!/usr/bin/env python
"""Is the chains' 2.2x tile stretch at full occupancy a POWER limit?  Same launches, same instruction stream, different operand data:
random (the benchmark's) against all-zero inputs and weights (minimal toggling in the MFMA / VALU / LDS / memory datapaths).  A
power-limited chip runs the zero arm markedly faster at 200 tiles and equally fast at 62; a resource-limited one does not care.
Also samples the board power (rocm-smi) while a long run of each arm is in flight.    python tools/power_probe.py"""
import ctypes as C
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H

lib = cva.lib.load()
cfg = O.NetConfig(input_dim=2048, hidden_dim=384, num_heads=8, ff_dim=384, pool_hidden=768, pool_heads=2)


def smi_power():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        return out.strip().replace("\n", " ")[:600]
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"


for arm in ("random", "zero"):
    P = O.make_params(cfg, 3)
    if arm == "zero":
        P = {k: np.zeros_like(v) for k, v in P.items()}
    net = H.make_hip_net(cfg, P, dropout=0.025)
    net.train(True)
    for N in (320, 200, 100):
        x = torch.randn(N, 80, 2048, device="cuda") if arm == "random" else torch.zeros(N, 80, 2048, device="cuda")
        lens = torch.full((N,), 80, dtype=torch.long, device="cuda")
        mask = torch.zeros(N, 80, dtype=torch.bool, device="cuda")
        with torch.no_grad():
            for _ in range(3):
                net(x, mask, lens, None, seed=1)
            torch.cuda.synchronize()
            lib.coot_timing_enable(1)
            for _ in range(10):
                net(x, mask, lens, None, seed=1)
            torch.cuda.synchronize()
            ms, fl, n = C.c_double(), C.c_double(), C.c_int()
            cva.lib.check(lib.coot_timing_collect(5, C.byref(ms), C.byref(fl), C.byref(n)), "timing_collect")
            lib.coot_timing_enable(0)
            print(f"[{arm}] N={N} ({N * 80 // 128} tiles): chain launches {1e3 * ms.value / 10:.1f} us per forward ({n.value // 10} launches)", flush=True)
            if N == 320:  # board power while ~3 s of back-to-back forwards are in flight
                samples = []
                stop = threading.Event()

                def poll():
                    while not stop.is_set():
                        samples.append(smi_power())
                        time.sleep(0.3)
                th = threading.Thread(target=poll)
                th.start()
                t0 = time.time()
                while time.time() - t0 < 4.0:
                    for _ in range(200):
                        net(x, mask, lens, None, seed=1)
                    torch.cuda.synchronize()
                stop.set()
                th.join()
                for s in samples[2:8]:
                    print(f"[{arm}]   smi: {s}", flush=True)
print("idle smi:", smi_power())
