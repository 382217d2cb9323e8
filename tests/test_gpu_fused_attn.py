"""SURVEY K3 (round 6): the forward self-attention of the local networks computed INSIDE post_attn_fwd_kernel (csrc/fused.hip:
fused_self_attn; coot_set_option("fused_attn", 0) switches it off) — for fixed-length sequences whose length is a multiple of 16 and
<= 80 (ActivityNet: 80 frames, 64 / 16 words) a token tile computes QK^T-softmax-PV for its own rows, wave = head, and the result lands
in the LDS tile of the output projection; no attn_short_fwd launch.  ON by default since its second version (1.197 against 1.210 ms per
step, profiles/r06_ab_fused_attn.txt), so the whole suite runs on it where the shapes allow; this file pins BOTH dispatches explicitly:
the reference's train-mode fixture (injected dropout masks: the probabilities' mask map must be the attention kernel's), the eval
fixture on both routes — with the option on (and the launch counter proving the path ran, and that shapes it does not cover fall back
per launch) and with it off (the attention kernels on the same fixtures).  The saved ctx / lse feed the unchanged attention backward
(gradient parity of the same tests)."""
import ctypes as C

import pytest

from tests import test_gpu_bench_parity as BP
from tests import test_gpu_train_parity as TP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import coot_videotext_amd as cva
    assert torch.cuda.is_available()
    cva.lib.load()
    return torch, cva


def _launches(cva):
    v = C.c_int(0)
    cva.lib.check(cva.lib.load().coot_get_option(b"fused_attn_launches", C.byref(v)), "coot_get_option")
    return v.value


@pytest.mark.parametrize("name", ["bench_anet_train", "bench_yc2_2d3d_2816_train"])
def test_attention_inside_the_chain_train_fixture(env, golden_dir, name):
    torch, cva = env
    lib = cva.lib.load()
    n0 = _launches(cva)
    lib.coot_set_option(b"fused_attn", 1)
    TP.test_train_mode_native_step_vs_reference_with_injected_masks(env, golden_dir, name)
    # anet: both local networks' forward (80-frame / 64- and 16-word sequences); yc2: Lc = 20 and Ls = 12 are no multiples of 16 —
    # those launches must have kept the separate attention kernel (and the test above still holds: the dispatch falls back per launch)
    got = _launches(cva) - n0
    assert (got >= 2) if name == "bench_anet_train" else (got == 0), got


def test_attention_inside_the_chain_eval_fixture_both_routes(env, golden_dir):
    torch, cva = env
    lib = cva.lib.load()
    n0 = _launches(cva)
    lib.coot_set_option(b"fused_attn", 1)
    BP.test_bench_shape_autograd_route(env, golden_dir, "bench_anet")
    BP.test_bench_shape_native_step(env, golden_dir, "bench_anet")
    assert _launches(cva) - n0 >= 4


def test_separate_attention_launches_still_pinned(env, golden_dir):
    """coot_set_option("fused_attn", 0): the same fixtures through the attention kernels (what shapes outside the in-chain path's
    reach, packed rows and the f16 / f32 builds' callers run)."""
    torch, cva = env
    lib = cva.lib.load()
    n0 = _launches(cva)
    lib.coot_set_option(b"fused_attn", 0)
    try:
        TP.test_train_mode_native_step_vs_reference_with_injected_masks(env, golden_dir, "bench_anet_train")
        BP.test_bench_shape_native_step(env, golden_dir, "bench_anet")
    finally:
        lib.coot_set_option(b"fused_attn", 1)
    assert _launches(cva) == n0
