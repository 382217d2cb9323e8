"""The drop-in boundary, exercised from the REFERENCE's side: the reference's own
RetrievalModelManager with the one added dispatch branch (coot_videotext_amd.integration.reference_manager_class) builds the HIP
networks from the reference's TransformerConfig objects, loads the state of the reference's TransformerLegacy networks strictly,
and keeps the BaseModelManager contract (get_all_params names and decay_mult, get / set_model_state incl. the legacy renames).

Container-only: needs /root/reference (skipped on the GPU box).  No GPU: the parameter layout is host code of the library."""
import copy
import os

import numpy as np
import pytest

from tests import helpers as H

REF = H.import_reference()
pytestmark = pytest.mark.skipif(REF is None, reason="the reference tree is not present (GPU box)")


def _ref_cfg(hip: bool):
    d = REF.utils_yaml.load_yaml_config_file(os.path.join(REF.root, "config/retrieval/paper2020/anet_coot.yaml"))
    d["use_cuda"] = False
    d["fp16_train"] = False
    d["fp16_val"] = False
    if hip:  # what `-o net_video_local.name=transformer_hip,...` does; the other three inherit through same_as or are set here
        for k in ("net_video_local", "net_video_global", "net_text_local", "net_text_global"):
            d[k]["name"] = "transformer_hip"
    return REF.configs_retrieval.RetrievalConfig(d)


def test_reference_manager_builds_hip_networks_and_keeps_the_manager_contract():
    import torch
    from coot_videotext_amd import integration
    from coot_videotext_amd.nets import TransformerHip
    torch.manual_seed(0)
    ref_mgr = REF.model_retrieval.RetrievalModelManager(_ref_cfg(False))       # four TransformerLegacy networks
    HipMgr = integration.reference_manager_class()
    assert issubclass(HipMgr, REF.model_retrieval.RetrievalModelManager)
    hip_mgr = HipMgr(_ref_cfg(True))
    assert list(hip_mgr.model_dict) == list(ref_mgr.model_dict)
    assert all(isinstance(m, TransformerHip) for m in hip_mgr.model_dict.values())
    # the adapter: the reference's config OBJECT (transformer_legacy.py:26-97) -> the fields the kernels take
    c = integration.hip_config_from_reference(_ref_cfg(True).model_cfgs["net_video_global"], 384)
    assert (c.hidden_dim, c.num_heads, c.ff_dim, c.use_context, c.use_input_fc, c.pooler, c.dropout, c.ctx_dropout) == \
        (384, 8, 384, True, False, "avg_special", 0.025, 0.025)
    c = integration.hip_config_from_reference(_ref_cfg(True).model_cfgs["net_text_local"], 1536)
    assert (c.input_dim, c.pooler, c.pool_hidden, c.pool_heads, c.pool_dropout) == (1536, "atn", 768, 2, 0.025)

    # 1. state dicts: same keys and shapes; the reference's state loads STRICTLY (load_state_dict default) and round-trips
    state = ref_mgr.get_model_state()
    assert not hip_mgr.was_loaded
    hip_mgr.set_model_state(copy.deepcopy(state))
    assert hip_mgr.was_loaded
    back = hip_mgr.get_model_state()
    for net in state:
        assert sorted(back[net].keys()) == sorted(state[net].keys()), net  # same entries (buffers included); the order is the module tree's
        for k in state[net]:
            assert torch.equal(back[net][k].cpu(), state[net][k]), (net, k)
    # ... with the "module." prefix of nn.DataParallel checkpoints too (utils_torch.edit_moduledot_in_state_keys)
    hip_mgr.set_model_state({net: {"module." + k: v for k, v in sd.items()} for net, sd in state.items()})

    # 2. legacy checkpoints of the first coot-videotext release: a LIST of state dicts with the old parameter names
    #    (model_manager_base.py:96-113 renames input_norm. / input_fc. / pooler.genpool)
    inverse = [("norm_input.", "input_norm."), ("input_fc.mlp.", "input_fc."), ("pooler.pools.0.genpool", "pooler.genpool")]
    legacy = []
    for net in state:
        sd = {}
        for k, v in state[net].items():
            for new, old in inverse:
                k = k.replace(new, old)
            sd[k] = v + 1.0  # different values, so that the load is visible
        legacy.append(sd)
    assert any("input_norm." in k for k in legacy[0]) and any("pooler.genpool" in k for k in legacy[0])
    hip_mgr.set_model_state(legacy)
    for net in state:
        for k, v in hip_mgr.model_dict[net].state_dict().items():
            assert torch.equal(v.cpu(), state[net][k] + 1.0), (net, k)

    # 3. optimizer-facing contract: names, count and decay_mult of get_all_params (bias decay_mult 0: weight_decay_for_bias)
    p_ref, n_ref, f_ref = ref_mgr.get_all_params()
    p_hip, n_hip, f_hip = hip_mgr.get_all_params()
    assert n_hip == n_ref and len(f_hip) == len(f_ref) == 118
    assert [p["decay_mult"] for p in p_hip] == [p["decay_mult"] for p in p_ref]
    assert [tuple(p["params"].shape) for p in p_hip] == [tuple(p["params"].shape) for p in p_ref]
    assert sum(int(np.prod(t.shape)) for t in f_hip if t.requires_grad) == 7604224
    # 4. train / eval switches reach the networks
    hip_mgr.set_all_models_eval()
    assert not hip_mgr.is_train and not any(m.training for m in hip_mgr.model_dict.values())
    hip_mgr.set_all_models_train()
    assert hip_mgr.is_train and all(m.training for m in hip_mgr.model_dict.values())


def test_own_manager_loads_legacy_list_checkpoints():
    """The standalone RetrievalModelManager honours the same legacy renames (weak point of round 1: only 'module.' was stripped)."""
    import torch
    import coot_videotext_amd as cva
    mgr = cva.RetrievalModelManager(cva.load_named_config("anet_coot"))
    state = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in mgr.model_dict.items()}
    inverse = [("norm_input.", "input_norm."), ("input_fc.mlp.", "input_fc."), ("pooler.pools.0.genpool", "pooler.genpool")]
    legacy = []
    for net in state:
        sd = {}
        for k, v in state[net].items():
            for new, old in inverse:
                k = k.replace(new, old)
            sd["module." + k] = v * 0.5
        legacy.append(sd)
    mgr.set_model_state(legacy)
    for net in state:
        for k, v in mgr.model_dict[net].state_dict().items():
            assert torch.equal(v, state[net][k] * 0.5), (net, k)


@pytest.mark.parametrize("max_violation,norm", [(True, True), (False, False), (True, False)])
def test_contrastive_loss_constructor_flags_match_the_reference_module(max_violation, norm):
    """ContrastiveLoss(margin, max_violation, norm) (coot/loss_fn.py:51-100): the two constructor flags no configuration reaches — the
    hardest negative per query only, the un-normalised sum — against the reference's module on the same normalised embeddings: loss and
    gradients wrt both inputs.  (The default flags are the fused HIP loss, pinned on the GPU by test_losses_vs_oracle and the fixtures.)"""
    import torch
    import coot_videotext_amd as cva
    from coot import loss_fn as ref_loss
    g = torch.Generator().manual_seed(3)
    base = torch.randn(1, 48, generator=g)
    im = torch.nn.functional.normalize(base + 0.7 * torch.randn(20, 48, generator=g), dim=-1)
    s = torch.nn.functional.normalize(im + 0.4 * torch.randn(20, 48, generator=g), dim=-1)
    outs = []
    for mod in (ref_loss.ContrastiveLoss(0.2, max_violation=max_violation, norm=norm, use_cuda=False),
                cva.ContrastiveLoss(0.2, max_violation=max_violation, norm=norm, use_cuda=False)):
        a, b = im.clone().requires_grad_(True), s.clone().requires_grad_(True)
        loss = mod(a, b)
        loss.backward()
        outs.append((float(loss), a.grad.clone(), b.grad.clone()))
    (lr, gar, gbr), (lo, gao, gbo) = outs
    assert lr > 0 and abs(lr - lo) <= 1e-6 * abs(lr)
    assert torch.allclose(gar, gao, rtol=1e-5, atol=1e-7) and torch.allclose(gbr, gbo, rtol=1e-5, atol=1e-7)
