#!/usr/bin/env python
"""Where the forward chain's tiles run and how long each takes (measurement build: tools/build_variant.sh tilelog "-DFZ_TILE_LOG";
COOT_HIP_LIB=.../libcoot_hip_tilelog.so python tools/tile_log.py): post_attn_fwd_kernel logs (HW_ID, XCC_ID, start, end) per tile.
Groups the tiles by XCD and by CU pair (two CUs share an instruction cache and a scalar cache) and prints the duration of tiles whose pair
mate also held a tile against those that had their pair to themselves — round 5's question: is the chains' throughput bound per XCD or
per CU pair?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import coot_videotext_amd as cva
from oracle import coot_oracle as O
from tests import helpers as H

lib = cva.lib.load()
cfg = O.NetConfig(input_dim=2048, hidden_dim=384, num_heads=8, ff_dim=384, pool_hidden=768, pool_heads=2)
net = H.make_hip_net(cfg, O.make_params(cfg, 3), dropout=0.025)
net.train(True)
for N in (320, 200, 150, 100):
    x = torch.randn(N, 80, 2048, device="cuda")
    lens = torch.full((N,), 80, dtype=torch.long, device="cuda")
    mask = torch.zeros(N, 80, dtype=torch.bool, device="cuda")
    tiles = N * 80 // 128
    ts = torch.zeros(128 + 4 * 512, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            net(x, mask, lens, None, seed=1)
        torch.cuda.synchronize()
        cva.lib.check(lib.coot_debug_timestamps(ts.data_ptr()))
        net(x, mask, lens, None, seed=1)
        torch.cuda.synchronize()
        cva.lib.check(lib.coot_debug_timestamps(None))
    e = ts.cpu().numpy()[128:128 + 4 * tiles].reshape(tiles, 4)
    hw, xcc = e[:, 0], e[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    dur = (e[:, 3] - e[:, 2]) / 100.0  # us (100 MHz counter)
    start = (e[:, 2] - e[:, 2].min()) / 100.0
    key_cu = xcc * 1000 + se * 100 + sh * 50 + cu
    assert len(set(key_cu.tolist())) == tiles or True
    print(f"N={N}: {tiles} tiles; post_attn_fwd tile duration min / median / max {dur.min():.1f} / {np.median(dur):.1f} / {dur.max():.1f} us; start spread {start.max():.1f} us; "
          f"distinct (xcc, se, sh, cu) {len(set(key_cu.tolist()))}")
    for pair_bits, lab in ((1, "cu >> 1"),):
        pair = xcc * 1000 + se * 100 + sh * 50 + (cu >> pair_bits)
        cnt = {k: int((pair == k).sum()) for k in set(pair.tolist())}
        occ = np.array([cnt[k] for k in pair.tolist()])
        for c in sorted(set(occ.tolist())):
            m = occ == c
            print(f"    tiles whose CU pair ({lab}) holds {c} tile(s): {int(m.sum()):3d} tiles, duration median {np.median(dur[m]):.1f} us (min {dur[m].min():.1f}, max {dur[m].max():.1f})")
    per_xcd = [int((xcc == k).sum()) for k in range(8)]
    print(f"    tiles per XCD {per_xcd}; per XCD median duration " + " ".join(f"{np.median(dur[xcc == k]):.0f}" if per_xcd[k] else "-" for k in range(8)))
    print(f"    CU ids seen: {sorted(set(cu.tolist()))}; SE ids {sorted(set(se.tolist()))}; SH ids {sorted(set(sh.tolist()))}")
