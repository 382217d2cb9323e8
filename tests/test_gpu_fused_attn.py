"""SURVEY K3, built in round 6 and OFF by default (it is not faster: profiles/r06_ab_fused_attn.txt): the forward self-attention of the
local networks computed INSIDE post_attn_fwd_kernel (csrc/fused.hip: fused_self_attn; coot_set_option("fused_attn", 1)) — for
fixed-length sequences whose length is a multiple of 16 (ActivityNet: 80 frames, 64 / 16 words) a token tile computes QK^T-softmax-PV
for its own rows and the result lands in the LDS tile of the output projection; no attn_short_fwd launch.  The path stays pinned:
the reference's train-mode fixture (injected dropout masks: the probabilities' mask map must be the attention kernel's), the eval
fixture on both routes, and the saved ctx / lse feed the unchanged attention backward (gradient parity of the same tests)."""
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
    try:
        TP.test_train_mode_native_step_vs_reference_with_injected_masks(env, golden_dir, name)
    finally:
        lib.coot_set_option(b"fused_attn", 0)
    # anet: both local networks' forward (80-frame / 64- and 16-word sequences); yc2: Lc = 20 and Ls = 12 are no multiples of 16 —
    # those launches must have kept the separate attention kernel (and the test above still holds: the dispatch falls back per launch)
    got = _launches(cva) - n0
    assert (got >= 2) if name == "bench_anet_train" else (got == 0), got


def test_attention_inside_the_chain_eval_fixture_both_routes(env, golden_dir):
    torch, cva = env
    lib = cva.lib.load()
    n0 = _launches(cva)
    lib.coot_set_option(b"fused_attn", 1)
    try:
        BP.test_bench_shape_autograd_route(env, golden_dir, "bench_anet")
        BP.test_bench_shape_native_step(env, golden_dir, "bench_anet")
    finally:
        lib.coot_set_option(b"fused_attn", 0)
    assert _launches(cva) - n0 >= 4
