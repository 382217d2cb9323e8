"""The REFERENCE'S OWN encode_visual / encode_text (coot/model_retrieval.py:86-197 — its python pack loop, its embedding tuples, its
batch class) running over TransformerHip networks on the GPU: the reference's RetrievalModelManager with the one dispatch branch of
INTEGRATION.md section 2 (coot_videotext_amd.integration.reference_manager_class), the reference's anet_coot.yaml with
`name: transformer_hip`, the reference's RetrievalDataBatchTuple — against the embeddings the unmodified reference (TransformerLegacy
on the CPU) wrote for the same parameters and batch (tests/golden/bench_anet.npz).

Needs the reference's python packages next to a GPU: skipped unless /root/reference or $COOT_REFERENCE_ROOT holds coot/ + nntrainer/ +
config/ (the GPU box has neither; for one run the three directories travel in a git-ignored copy, profiles/r04_reference_on_gpu.log)."""
import os

import numpy as np
import pytest

from oracle import coot_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

REF = H.import_reference()


@pytest.mark.skipif(REF is None, reason="the reference tree is not present")
@pytest.mark.parametrize("name", ["bench_anet", "bench_anet_ragged"])
def test_reference_encode_visual_and_text_over_hip_networks(golden_dir, name):
    import torch
    from coot import dataset_retrieval as ref_ds
    from coot_videotext_amd import integration
    from coot_videotext_amd.nets import TransformerHip
    assert torch.cuda.is_available()
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    seed, B, Lv, Lc, Lp, Ls, dv, dt, hidden, heads, ff, ph = [int(v) for v in g["meta"]]
    cfgs = H.full_cfgs(dv, dt, hidden, heads, ff, ph)
    Ps = [O.make_params(cfgs[i], seed + 10 * i, scale=float(g["param_scale"])) for i in range(4)]
    b = O.make_batch(seed + 100, B, g["counts"], Lv, Lc, Lp, Ls, dv, dt, ragged=bool(int(g["ragged"])), corr=0.5)

    d = REF.utils_yaml.load_yaml_config_file(os.path.join(REF.root, "config/retrieval/paper2020/anet_coot.yaml"))
    d["use_cuda"], d["fp16_train"], d["fp16_val"] = True, False, False
    for k in H.NET_KEYS:
        d[k]["name"] = "transformer_hip"
    cfg = REF.configs_retrieval.RetrievalConfig(d)
    mgr = integration.reference_manager_class()(cfg)                 # the reference's manager class, HIP networks inside
    assert isinstance(mgr, REF.model_retrieval.RetrievalModelManager)
    for k, P in zip(H.NET_KEYS, Ps):
        net = mgr.model_dict[k]
        assert isinstance(net, TransformerHip)
        sd = net.state_dict()
        for n in sd:
            if n in P:
                sd[n] = torch.from_numpy(np.asarray(P[n], dtype=np.float32))
        net.load_state_dict(sd)
        net.cuda()
    mgr.set_all_models_eval()

    keys = [str(i) for i in range(B)]
    t = {k: torch.as_tensor(np.asarray(v)).cuda() for k, v in b.items()}
    batch = ref_ds.RetrievalDataBatchTuple(                          # the reference's own batch class (typed, shape-validated)
        key=keys, data_key=keys, sentences=[["w"]] * B, vid_feat=t["vid_feat"].float(), vid_feat_mask=t["vid_feat_mask"], vid_feat_len=t["vid_feat_len"],
        par_feat=t["par_feat"].float(), par_feat_mask=t["par_feat_mask"], par_feat_len=t["par_feat_len"], clip_num=t["clip_num"],
        clip_feat=t["clip_feat"].float(), clip_feat_mask=t["clip_feat_mask"], clip_feat_len=t["clip_feat_len"], sent_num=t["sent_num"],
        sent_feat=t["sent_feat"].float(), sent_feat_mask=t["sent_feat_mask"], sent_feat_len=t["sent_feat_len"])
    with torch.no_grad():
        vis = mgr.encode_visual(batch)                               # coot/model_retrieval.py:86-141, unmodified
        txt = mgr.encode_text(batch)                                 # coot/model_retrieval.py:143-197
    torch.cuda.synchronize()
    assert type(vis).__module__.startswith("coot.") and type(txt).__module__.startswith("coot.")
    got = {"vid_emb": vis.vid_emb, "clip_emb": vis.clip_emb, "vid_context": vis.vid_context, "par_emb": txt.par_emb, "sent_emb": txt.sent_emb,
           "par_context": txt.par_context}
    for k, v in got.items():
        cos = H.cosine_rows(v.cpu().numpy(), g[k]).min()
        print(f"[{name}] reference encode_* over TransformerHip: {k} min row cosine vs the reference's own networks {cos:.6f}")
        assert cos > 1 - 1e-3, (k, cos)
    # the reference's python pack loop produced the same packed tensor / mask / lengths as the fixture
    assert np.array_equal(vis.clip_emb_lens.cpu().numpy(), g["clip_emb_lens"]) and np.array_equal(vis.clip_emb_mask.cpu().numpy(), g["clip_emb_mask"].astype(bool))
