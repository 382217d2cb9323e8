"""Gradient WRITE mode under NaN poison (coot_net_grads_overwrite / coot_set_option("grad_write"), csrc/api_step.hip).

coot_train_step does not zero the weight-matrix gradients: its backward writes them, each being the result of exactly one
weight-gradient problem.  A code path that still ACCUMULATES into one of them would add onto the previous step's values without
anyone noticing.  coot_set_option("grad_poison", 1) fills every matrix the zero launch skips with NaN first: any accumulate
shows up as NaN.  Paths: fused chains (d_model 384, >= 1024 tokens), per-op kernels ("fused" = 0), packed token rows, a side
below the fused threshold (small T), two encoder layers (per-layer flushes); single call and the data-parallel phase calls.
Reference behaviour being replaced: optimizer.zero_grad() + autograd accumulation (coot/trainer_retrieval.py:276-283).
"""
import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    return torch, cva


CASES = [  # name, options, packed, layers
    ("fused", {}, False, 1), ("per-op", {"fused": 0}, False, 1), ("packed", {}, True, 1), ("two-layer", {}, False, 2),
    ("per-op global passes", {"glob_fused": 0, "glob_fused_bwd": 0}, False, 1),
]


@pytest.mark.parametrize("name,opts,packed,layers", CASES)
def test_written_gradients_under_nan_poison_equal_fill_and_accumulate(env, name, opts, packed, layers):
    torch, cva = env
    lib = cva.lib.load()
    dims = (256, 128, 384, 8, 384, 768)
    cfgs = H.full_cfgs(*dims, layers=layers)
    Ps = [O.make_params(cfgs[i], 3 + i, scale=0.05) for i in range(4)]
    counts = [3, 1, 2, 4, 2, 3, 1, 2]
    # video side: 8 x 80 + 18 x 64 = 1 792 tokens (fused chains); text side: 8 x 24 + 18 x 12 = 408 tokens (below the fused threshold)
    b = O.make_batch(17, 8, counts, 80, 64, 24, 12, dims[0], dims[1], ragged=True, corr=0.5)
    idx = torch.zeros(16, dtype=torch.int64, device="cuda")
    grads = {}
    try:
        for k, v in opts.items():
            cva.lib.check(lib.coot_set_option(k.encode(), v), k)
        for mode in ("write+poison", "fill+accumulate"):
            cva.lib.check(lib.coot_set_option(b"grad_write", 1 if mode == "write+poison" else 0), "grad_write")
            cva.lib.check(lib.coot_set_option(b"grad_poison", 1 if mode == "write+poison" else 0), "grad_poison")
            cfg, mgr = H.make_manager(cfgs, Ps, dropout=0.0, cc_weight=0.01)
            mgr.set_all_models_train()
            tr = cva.RetrievalTrainer(cfg, mgr)
            batch = cva.synthetic.batch_from_numpy(b, packed=packed)
            for _ in range(2):  # the second step meets the first one's gradients in the arenas
                tr.train_step_native(batch, do_optimizer=False, cc_indices=idx)
            torch.cuda.synchronize()
            grads[mode] = [n._grad_flat.detach().cpu().numpy().copy() for n in mgr.model_dict.values()]
    finally:
        for k in opts:
            lib.coot_set_option(k.encode(), 1)
        lib.coot_set_option(b"grad_write", 1)
        lib.coot_set_option(b"grad_poison", 0)
    for key, a, c in zip(H.NET_KEYS, grads["write+poison"], grads["fill+accumulate"]):
        assert np.isfinite(a).all(), (name, key, int((~np.isfinite(a)).sum()))
        scale = float(np.abs(c).max())
        err = float(np.abs(a - c).max()) / scale
        assert err < 2e-4, (name, key, err)  # fp32 summation order of the split reduction only
