"""
Golden-vector generator: runs the UNMODIFIED reference (imported from /root/reference with the
import shims of SURVEY.md section 8c) on seeded inputs and writes tests/golden/*.npz.

Runs only in the build container (the reference tree does not travel to the GPU box).  The
committed fixtures pin oracle/coot_oracle.py (tests/test_oracle_golden.py); the HIP path is then
checked against the oracle and, for the end-to-end cases, directly against these fixtures.

    python oracle/gen_golden.py            # regenerate everything
"""
import collections
import collections.abc
import copy
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("COOT_REFERENCE", "/root/reference")
OUT = os.environ.get("COOT_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))  # (tests regenerate into a scratch directory and compare)

# ---- import shims (no reference edits) -------------------------------------------------------
for _n in ("Iterable", "Mapping", "Sequence", "MutableMapping"):
    setattr(collections, _n, getattr(collections.abc, _n))
for _name in ("GPUtil", "h5py"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
_tb = types.ModuleType("torch.utils.tensorboard")  # nntrainer/metric.py:17 (tensorboard absent here)
_tb.SummaryWriter = type("SummaryWriter", (), {"add_scalar": lambda *a, **k: None, "close": lambda *a, **k: None})
sys.modules.setdefault("torch.utils.tensorboard", _tb)
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import torch as th  # noqa: E402
from torch.nn import functional as F  # noqa: E402

from coot import model_retrieval  # noqa: E402
from coot.configs_retrieval import RetrievalConfig  # noqa: E402
from coot.dataset_retrieval import RetrievalDataBatchTuple  # noqa: E402
from coot.loss_fn import ContrastiveLoss, CycleConsistencyLoss  # noqa: E402
from coot.trainer_retrieval import RetrievalTrainer  # noqa: E402
from nntrainer import models, retrieval, utils_yaml  # noqa: E402

from oracle import coot_oracle as O  # noqa: E402

th.set_num_threads(4)


def ref_config(dv, dt, hidden, heads, ff, pool_hidden, layers=1, dropout=None):
    """The reference's ActivityNet config with other widths.  layers: encoder layers of the LOCAL networks (the global networks
    copy the local section through same_as, anet_coot.yaml:111-113, so their own self-attention section is spelled out to keep
    them at one layer: 'YouCook2-100m d=512, 2-layer local/1-layer global' in BASELINE.json).  dropout: every dropout
    probability of the four networks (self-attention, context and pooler sections)."""
    d = utils_yaml.load_yaml_config_file(os.path.join(REF, "config/retrieval/paper2020/anet_coot.yaml"))
    d["use_cuda"] = False
    d["fp16_train"] = False
    d["fp16_val"] = False
    d["dataset_train"]["vid_feat_dim"] = dv
    d["dataset_train"]["text_feat_dim"] = dt
    loc = d["net_video_local"]
    loc["output_dim"] = hidden
    loc["input_fc_config"]["output_dim"] = hidden
    loc["selfatn_config"].update(hidden_dim=hidden, num_heads=heads, pointwise_ff_dim=ff, num_layers=layers)
    loc["pooler_config"]["hidden_dim"] = pool_hidden
    glob = d["net_video_global"]
    glob["output_dim"] = 2 * hidden
    glob["crossatn_config"].update(hidden_dim=hidden, num_heads=heads, pointwise_ff_dim=ff)
    if layers != 1:
        glob["selfatn_config"] = dict(loc["selfatn_config"], num_layers=1)
    if dropout is not None:
        loc["selfatn_config"]["dropout"] = dropout
        loc["pooler_config"]["dropout"] = dropout
        glob["crossatn_config"]["dropout"] = dropout
        if "selfatn_config" in glob:
            glob["selfatn_config"]["dropout"] = dropout
    return RetrievalConfig(d)


def oracle_cfgs(dv, dt, hidden, heads, ff, pool_hidden, layers=1):
    kw = dict(hidden_dim=hidden, num_heads=heads, ff_dim=ff, pool_hidden=pool_hidden)
    loc_v = O.NetConfig(input_dim=dv, num_layers=layers, **kw)
    loc_t = O.NetConfig(input_dim=dt, num_layers=layers, **kw)
    glob = O.NetConfig(input_dim=hidden, use_input_fc=False, use_context=True, pooler="avg_special", **kw)
    return loc_v, glob, loc_t, copy.deepcopy(glob)


NET_KEYS = ["net_video_local", "net_video_global", "net_text_local", "net_text_global"]


def load_params(module, P):
    sd = module.state_dict()
    for k in sd:
        if k in P:
            sd[k] = th.from_numpy(np.asarray(P[k], dtype=np.float32))
    module.load_state_dict(sd)


def to_batch(b):
    t = {k: th.from_numpy(np.asarray(v)) for k, v in b.items()}
    B = len(b["clip_num"])
    f = lambda k: t[k].float()
    return RetrievalDataBatchTuple(
        key=[str(i) for i in range(B)], data_key=[str(i) for i in range(B)], sentences=[[""]] * B,
        vid_feat=f("vid_feat"), vid_feat_mask=t["vid_feat_mask"], vid_feat_len=t["vid_feat_len"],
        par_feat=f("par_feat"), par_feat_mask=t["par_feat_mask"], par_feat_len=t["par_feat_len"],
        clip_num=t["clip_num"], clip_feat=f("clip_feat"), clip_feat_mask=t["clip_feat_mask"],
        clip_feat_len=t["clip_feat_len"], sent_num=t["sent_num"], sent_feat=f("sent_feat"),
        sent_feat_mask=t["sent_feat_mask"], sent_feat_len=t["sent_feat_len"])


class _FakeTrainer:
    """Just enough of RetrievalTrainer to call its loss hooks unbound (no dirs/loggers)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.loss_contr = ContrastiveLoss(cfg.train.contrastive_loss_config.margin, use_cuda=False)
        self.loss_cycle_cons = CycleConsistencyLoss(num_samples=1, use_cuda=False)

    compute_align_loss = RetrievalTrainer.compute_align_loss
    compute_cluster_loss = RetrievalTrainer.compute_cluster_loss
    compute_total_constrastive_loss = RetrievalTrainer.compute_total_constrastive_loss
    compute_cyclecons_loss = RetrievalTrainer.compute_cyclecons_loss


def draw_cc_indices(seed, clip_mask, sent_mask):
    """Replays the RNG consumption order of CycleConsistencyLoss.get_total_loss
    (coot/loss_fn.py:306-314): B draws for the clip cycle, then B for the sentence cycle."""
    th.manual_seed(seed)
    ic = [int(th.multinomial((~m).float(), 1)) for m in clip_mask]
    is_ = [int(th.multinomial((~m).float(), 1)) for m in sent_mask]
    return np.array(ic), np.array(is_)


def subsample(a, step=97):
    return np.asarray(a, dtype=np.float32).reshape(-1)[::step].copy()


# ---- train-mode parity: the library's dropout masks injected into the unmodified reference ---------------------------------
from oracle import dropout_masks as DM  # noqa: E402
import re  # noqa: E402

_SITE_PATTERNS = [  # module path inside a TransformerLegacy -> (kind, site offset); group(1) = layer index where there is one
    (re.compile(r"^tf\.encoder_layers\.(\d+)\.self_attention_layer\.sublayer\.dropout$"), "attn", DM.SITE_ATTN, 0),
    (re.compile(r"^tf\.encoder_layers\.(\d+)\.dropout$"), "rows", DM.SITE_POSTLN, 0),
    (re.compile(r"^tf\.encoder_layers\.(\d+)\.pointwise_feedforward_layer\.sublayer\.feed_forward\.1$"), "rows", DM.SITE_FF1, 0),
    (re.compile(r"^tf\.encoder_layers\.(\d+)\.pointwise_feedforward_layer\.sublayer\.feed_forward\.4$"), "rows", DM.SITE_FF2, 0),
    (re.compile(r"^tf_context\.encoder_layers\.(\d+)\.self_attention_layer\.sublayer\.dropout$"), "cattn", DM.SITE_ATTN, 8),
    (re.compile(r"^tf_context\.encoder_layers\.(\d+)\.dropout$"), "crows", DM.SITE_POSTLN, 8),
    (re.compile(r"^tf_context\.encoder_layers\.(\d+)\.pointwise_feedforward_layer\.sublayer\.feed_forward\.1$"), "crows", DM.SITE_FF1, 8),
    (re.compile(r"^tf_context\.encoder_layers\.(\d+)\.pointwise_feedforward_layer\.sublayer\.feed_forward\.4$"), "crows", DM.SITE_FF2, 8),
    (re.compile(r"^pooler\.pools\.0\.dropout1()$"), "pool", DM.SITE_POOL1, 15),
    (re.compile(r"^pooler\.pools\.0\.dropout2()$"), "pool", DM.SITE_POOL2, 15),
    (re.compile(r"^pooler\.pools\.0\.dropout3()$"), "pool3", DM.SITE_POOL3, 15),
]


class _InjectedDropout(th.nn.Module):
    """Stands where an nn.Dropout stood: multiplies with the keep-scales the HIP library draws for this site and this call."""

    def __init__(self, state, kind, site, p):
        super().__init__()
        self.state, self.kind, self.site, self.p = state, kind, site, p

    def forward(self, x):
        lay, seed, shp = self.state["layout"], self.state["seed"], tuple(x.shape)
        if self.kind == "rows":       # [N, L, ld]
            assert shp[:2] == (lay.N, lay.L), (shp, lay.N, lay.L)
            m = DM.mask_rows(seed, self.site, lay, shp[2], self.p)
        elif self.kind == "crows":    # context layer: one query row per sequence [N, 1, ld]
            assert shp[:2] == (lay.N, 1)
            m = DM.mask_rows(seed, self.site, DM.CallLayout(0, lay.N, 1, 0), shp[2], self.p)
        elif self.kind == "attn":     # [N, H, L, L]
            assert shp[0] == lay.N and shp[2] == shp[3] == lay.L
            m = DM.mask_attention(seed, self.site, lay, shp[1], lay.L, lay.L, self.p)
        elif self.kind == "cattn":    # [N, H, 1, L]
            assert shp[0] == lay.N and shp[2] == 1 and shp[3] == lay.L
            m = DM.mask_attention(seed, self.site, DM.CallLayout(0, lay.N, lay.L, 0), shp[1], 1, lay.L, self.p)
        else:                         # GenPool: [N, heads, L, d]
            assert shp[0] == lay.N and shp[2] == lay.L
            m = DM.mask_heads(seed, self.site, lay, shp[1], shp[3], self.p, pool3=(self.kind == "pool3"))
        self.state["used"].add(self.site)
        return x * th.from_numpy(np.ascontiguousarray(m)).reshape(shp)


def inject_dropout(net, seed, p, packed=False):
    """Replaces every active nn.Dropout of one reference TransformerLegacy (nntrainer/models/transformer_legacy.py:418,435,487,
    553,592-598; nntrainer/models/poolers.py:139-143) by _InjectedDropout; a forward pre-hook keeps track of which SEGMENT of
    the library's network call the current module call is (the reference calls a local network twice per step, on the videos
    and on the clips: coot/model_retrieval.py:104,120; the library runs both through one call).  Returns the shared state."""
    state = dict(seed=seed, calls=0, row_next=0, tok_next=0, layout=None, used=set(), packed=packed)
    todo = []
    for name, mod in net.named_modules():
        if not isinstance(mod, th.nn.Dropout):
            continue
        for pat, kind, off, base in _SITE_PATTERNS:
            mt = pat.match(name)
            if mt:
                layer = int(mt.group(1)) if mt.group(1) else 0
                todo.append((name, _InjectedDropout(state, kind, 16 * (base + layer) + off, p)))
                break
        else:
            assert mod.p == 0, f"dropout module {name} (p = {mod.p}) has no site in the library"
    for name, new in todo:
        parent = net
        parts = name.split(".")
        for a in parts[:-1]:
            parent = getattr(parent, a)
        setattr(parent, parts[-1], new)

    def pre_hook(_mod, args):
        feats, _mask, lengths = args[0], args[1], args[2]
        N, L = int(feats.shape[0]), int(feats.shape[1])
        lens = lengths.numpy().astype(np.int64)
        seg = state["calls"]
        assert seg < 2
        cu = None
        if state["packed"]:
            cu = state["tok_next"] + np.concatenate([[0], np.cumsum(lens)[:-1]])
        state["layout"] = DM.CallLayout(seg, N, L, state["row_next"], lens, cu)
        state["calls"] += 1
        state["row_next"] += N * L
        state["tok_next"] += int(lens.sum())

    net.register_forward_pre_hook(pre_hook)
    return state


def gen_single_net(name, cfg_o: O.NetConfig, N, L, seed, with_ctx):
    """One TransformerLegacy: forward (pooled, per-token) + grads of sum(pooled * R)."""
    tc = dict(name="transformer", output_dim=cfg_o.hidden_dim * (2 if with_ctx else 1), dropout_input=0,
              norm_input="layernorm_coot", positional_encoding="sincos", add_local_cls_token=False,
              use_input_fc=cfg_o.use_input_fc,
              selfatn_config=dict(hidden_dim=cfg_o.hidden_dim, num_layers=cfg_o.num_layers, dropout=0.025,
                                  num_heads=cfg_o.num_heads, pointwise_ff_dim=cfg_o.ff_dim, activation="gelu",
                                  norm="layernorm_coot"),
              use_output_fc=False, use_context=with_ctx,
              pooler_config=(dict(name="atn", hidden_dim=cfg_o.pool_hidden, num_heads=cfg_o.pool_heads,
                                  num_layers=1, dropout=0.025, activation="gelu") if cfg_o.pooler == "atn"
                             else dict(name="avg_special")),
              weight_init_type="truncnorm", weight_init_std=0.01)
    if cfg_o.use_input_fc:
        tc["input_fc_config"] = dict(output_dim=cfg_o.hidden_dim, num_layers=1, hidden_dim=0,
                                     activation_middle="none", activation_output="gelu", dropout_middle=0,
                                     dropout_output=0, norm_middle="none", norm_output="none", residual="none")
    if with_ctx:
        tc["crossatn_config"] = dict(hidden_dim=cfg_o.hidden_dim, num_layers=cfg_o.ctx_num_layers, dropout=0.025,
                                     num_heads=cfg_o.num_heads, pointwise_ff_dim=cfg_o.ff_dim,
                                     activation="gelu", norm="layernorm_coot")
    net = models.TransformerLegacy(models.TransformerConfig(tc), cfg_o.input_dim)
    P = O.make_params(cfg_o, seed)
    load_params(net, P)
    net.eval()
    rs = np.random.RandomState(seed + 1)
    lens = rs.randint(1, L + 1, size=N)
    lens[0] = L
    x = rs.randn(N, L, cfg_o.input_dim)
    x[np.arange(L)[None, :] >= lens[:, None]] = 0
    hid = rs.randn(N, cfg_o.hidden_dim) if with_ctx else None
    R = rs.randn(N, cfg_o.hidden_dim * (2 if with_ctx else 1))
    xt = th.from_numpy(x).float().requires_grad_(True)
    ht = th.from_numpy(hid).float().requires_grad_(True) if with_ctx else None
    mask = th.from_numpy(np.arange(L)[None, :] >= lens[:, None])
    pooled, per_tok = net(xt, mask, th.from_numpy(lens), ht)
    (pooled * th.from_numpy(R).float()).sum().backward()
    out = dict(x=x.astype(np.float32), lens=lens, R=R.astype(np.float32), pooled=pooled.detach().numpy(),
               per_token=per_tok.detach().numpy(), dx=xt.grad.numpy())
    if with_ctx:
        out["hidden"] = hid.astype(np.float32)
        out["dhidden"] = ht.grad.numpy()
    for k, p in net.named_parameters():
        if p.grad is not None:
            out["grad:" + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if not k.startswith("grad:")})


def gen_full(name, dims, B, counts, Ls, seed, full_grads, ragged=True, cc_weight=None, sub_step=97, scale=0.05,
             store_reshape=True, layers=1, train=None):
    """encode_visual + encode_text + total contrastive + cycle-consistency, fwd and bwd.

    train = dict(p=..., step_seed=..., packed=bool): TRAIN mode with every dropout site active at probability p, the masks being
    the ones the HIP library draws for coot_train_step(seed = step_seed) (oracle/dropout_masks.py; packed: the masks of the
    packed token-row layout).  Otherwise eval mode (dropout off)."""
    dv, dt, hidden, heads, ff, pool_hidden = dims
    Lv, Lc, Lp, Lsent = Ls
    cfg = ref_config(*dims, layers=layers, dropout=(train["p"] if train else None))
    if cc_weight is not None:
        cfg.train.loss_cycle_cons = cc_weight
    ocfgs = oracle_cfgs(*dims, layers=layers)
    th.manual_seed(0)
    mgr = model_retrieval.RetrievalModelManager(cfg)
    for i, k in enumerate(NET_KEYS):
        load_params(mgr.model_dict[k], O.make_params(ocfgs[i], seed + 10 * i, scale=scale))
    mgr.set_all_models_eval()
    drop_states = []
    if train:
        mgr.set_all_models_train()
        for k, net_seed in zip(NET_KEYS, DM.step_net_seeds(int(train["step_seed"]))):
            drop_states.append(inject_dropout(mgr.model_dict[k], net_seed, float(train["p"]),
                                              packed=bool(train.get("packed")) and k.endswith("local")))
    b = O.make_batch(seed + 100, B, counts, Lv, Lc, Lp, Lsent, dv, dt, ragged=ragged, corr=0.5)
    batch = to_batch(b)
    vis = mgr.encode_visual(batch)
    txt = mgr.encode_text(batch)
    tr = _FakeTrainer(cfg)
    contr = tr.compute_total_constrastive_loss(vis, txt)
    ic, isent = draw_cc_indices(seed + 7, vis.clip_emb_mask, txt.sent_emb_mask)
    th.manual_seed(seed + 7)
    cc = tr.compute_cyclecons_loss(vis, txt)
    loss = contr + cc
    loss.backward()
    # full per-position cycle losses (deterministic comparison target)
    with th.no_grad():
        ccl = tr.loss_cycle_cons
        cm, sm = ~vis.clip_emb_mask, ~txt.sent_emb_mask
        nn1, _, _ = ccl.get_soft_nn(vis.clip_emb_reshape, cm, txt.sent_emb_reshape, sm)
        _, beta, _ = ccl.get_soft_nn(nn1, cm, vis.clip_emb_reshape, cm)
        lrow_c, _, _ = ccl.compute_loss_index_gauss(cm, None, cm.shape[1], beta)
        nn2, _, _ = ccl.get_soft_nn(txt.sent_emb_reshape, sm, vis.clip_emb_reshape, cm)
        _, beta2, _ = ccl.get_soft_nn(nn2, sm, txt.sent_emb_reshape, sm)
        lrow_s, _, _ = ccl.compute_loss_index_gauss(sm, None, sm.shape[1], beta2)
    out = dict(vid_emb=vis.vid_emb, clip_emb=vis.clip_emb, vid_context=vis.vid_context,
               clip_emb_reshape=vis.clip_emb_reshape, clip_emb_mask=vis.clip_emb_mask,
               clip_emb_lens=vis.clip_emb_lens, par_emb=txt.par_emb, sent_emb=txt.sent_emb,
               par_context=txt.par_context, sent_emb_reshape=txt.sent_emb_reshape,
               sent_emb_mask=txt.sent_emb_mask, sent_emb_lens=txt.sent_emb_lens,
               contr_loss=contr, cc_loss=cc, cc_rows_clip=lrow_c, cc_rows_sent=lrow_s)
    out = {k: v.detach().numpy() for k, v in out.items()}
    if not store_reshape:  # the padded [B, Cmax, D] copies are pack_by_count(clip_emb / sent_emb): derivable, large
        del out["clip_emb_reshape"], out["sent_emb_reshape"]
    out["cc_idx_clip"], out["cc_idx_sent"] = ic, isent
    out["meta"] = np.array([seed, B, Lv, Lc, Lp, Lsent, dv, dt, hidden, heads, ff, pool_hidden])
    out["ragged"] = np.array(int(ragged))
    out["cc_weight"] = np.array(float(cfg.train.loss_cycle_cons))
    out["sub_step"] = np.array(sub_step)
    out["param_scale"] = np.array(scale)
    out["counts"] = np.asarray(counts)
    out["layers"] = np.array(layers)
    if train:
        for k, stt in zip(NET_KEYS, drop_states):  # every site of the library's table was hit (7 per local, 8 per global network)
            want = 7 * layers - 3 * (layers - 1) if k.endswith("local") else 8
            assert len(stt["used"]) == want, (k, sorted(stt["used"]))
        out["train_p"] = np.array(float(train["p"]))
        out["train_step_seed"] = np.array(int(train["step_seed"]), dtype=np.uint64)
        out["train_packed"] = np.array(int(bool(train.get("packed"))))
    for k in NET_KEYS:
        for n, p in mgr.model_dict[k].named_parameters():
            if p.grad is None:
                continue
            g = p.grad.numpy()
            if full_grads:
                out[f"grad:{k}:{n}"] = g
            else:
                out[f"gnorm:{k}:{n}"] = np.array(np.linalg.norm(g.astype(np.float64)))
                # vectors (biases, LayerNorm parameters) in full: a stride-197 sample of 384 numbers is two numbers
                out[f"gsub:{k}:{n}"] = subsample(g, sub_step if g.size > 4096 else 1)
    # retrieval metrics on these embeddings (nntrainer/retrieval.py)
    for (a, c, tag) in (("vid_emb", "par_emb", "vp"), ("clip_emb", "sent_emb", "cs")):
        e1 = out[a] / np.sqrt((out[a] ** 2).sum(-1, keepdims=True))
        e2 = out[c] / np.sqrt((out[c] ** 2).sum(-1, keepdims=True))
        r1, r2, s1, _ = retrieval.compute_retrieval({"a": e1, "b": e2}, "a", "b", print_fn=lambda *_: None)
        out[f"ret_{tag}"] = np.array([r1[k] for k in ("r1", "r5", "r10", "r50", "medr", "meanr")] +
                                     [r2[k] for k in ("r1", "r5", "r10", "r50", "medr", "meanr")] + [s1])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "contr", float(contr), "cc", float(cc))


def gen_retrieval_metrics():
    rs = np.random.RandomState(5)
    out = {}
    for i, n in enumerate((7, 64, 301)):
        d = rs.randn(n, n).astype(np.float32)
        d[np.arange(n), np.arange(n)] += 1.5
        if i == 1:  # force ties, incl. ties with the diagonal (argsort()[::-1] tie order matters)
            d = np.round(d * 2) / 2
        res, top1, ranks = retrieval.compute_retrieval_cosine(d)
        out[f"d{i}"] = d
        out[f"ranks{i}"] = ranks
        out[f"res{i}"] = np.array([res[k] for k in ("r1", "r5", "r10", "r50", "medr", "meanr")])
    np.savez_compressed(os.path.join(OUT, "retrieval_metrics.npz"), **out)
    print("wrote retrieval_metrics")


def gen_radam():
    """The reference's in-file RAdam (nntrainer/optimization.py:79-181) for 9 steps from a seeded state, with and without
    degenerated_to_sgd, two parameter groups (weight_decay, and decay_mult = 0): parameters after every step."""
    from nntrainer import optimization
    rs = np.random.RandomState(11)
    n = 257
    p0 = (rs.randn(n) * 0.05).astype(np.float32)
    grads = [(rs.randn(n) * (0.1 ** (s % 3))).astype(np.float32) for s in range(9)]
    out = {"p0": p0, "grads": np.stack(grads), "lr": 3.6e-4, "beta1": 0.56, "beta2": 0.98, "eps": 1.5e-9, "wd": 2e-3}
    for degen in (False, True):
        pa = th.nn.Parameter(th.from_numpy(p0[:200].copy()))
        pb = th.nn.Parameter(th.from_numpy(p0[200:].copy()))
        opt = optimization.RAdam([dict(params=[pa], weight_decay=2e-3), dict(params=[pb], weight_decay=0.0)], lr=3.6e-4,
                                 betas=(0.56, 0.98), eps=1.5e-9, degenerated_to_sgd=degen)
        traj = []
        for g in grads:
            pa.grad = th.from_numpy(g[:200].copy()); pb.grad = th.from_numpy(g[200:].copy())
            opt.step()
            traj.append(np.concatenate([pa.detach().numpy(), pb.detach().numpy()]))
        out[f"traj_degen{int(degen)}"] = np.stack(traj)
    np.savez_compressed(os.path.join(OUT, "radam.npz"), **out)
    print("wrote radam")


LR_SCHEDULE_CASES = [  # (name, scheduler config, steps per epoch, epochs)
    ("anet", dict(name="reduce_opw", warmup_type="epoch", warmup_epochs=3, rop_factor=0.1, rop_patience=2, rop_cooldown=3,
                  rop_min_lr_factor=0), 5, 40),
    ("yc2", dict(name="reduce_opw", warmup_type="epoch", warmup_epochs=0, rop_factor=0.1, rop_patience=5, rop_cooldown=3,
                 rop_min_lr_factor=0), 3, 60),
    ("stepwarm", dict(name="reduce_opw", warmup_type="step", warmup_epochs=2, rop_factor=0.5, rop_patience=1, rop_cooldown=1,
                      rop_min_lr_factor=0.01), 7, 50),
    ("nowarm", dict(name="reduce_opw", warmup_type="none", warmup_epochs=4, rop_factor=0.3, rop_patience=0, rop_cooldown=0,
                    rop_min_lr_factor=0.05), 2, 30),
    ("const", dict(name="none", warmup_type="step", warmup_epochs=3), 4, 8),
]


def lr_schedule_flags(case_idx, epochs):
    """Per epoch: (validated?, new best?) — drawn so that plateaus of every length up to 8 occur."""
    rs = np.random.RandomState(100 + case_idx)
    is_val = rs.rand(epochs) < 0.85
    improved = rs.rand(epochs) < 0.3
    return is_val, improved & is_val


def gen_lr_schedule():
    """nntrainer/lr_scheduler.py driven the way the trainer drives it (step() per train step, step_epoch() per epoch): the
    learning rate of both parameter groups and current_lr after EVERY call, for the shipped ANet / YC2 schedules and three
    more covering per-step warmup, no warmup, a minimum LR factor and the constant schedule."""
    from nntrainer import lr_scheduler
    out = {}
    for ci, (name, sc, spe, epochs) in enumerate(LR_SCHEDULE_CASES):
        pa, pb = th.nn.Parameter(th.zeros(3)), th.nn.Parameter(th.zeros(2))
        opt = th.optim.Adam([dict(params=[pa], lr=1e-3), dict(params=[pb], lr=2.5e-4)], lr=1e-3)
        sched = lr_scheduler.make_lr_scheduler(opt, lr_scheduler.SchedulerConfig(dict(sc)), 1e-3, epochs, spe)
        is_val, improved = lr_schedule_flags(ci, epochs)
        rows = [[opt.param_groups[0]["lr"], opt.param_groups[1]["lr"], sched.current_lr]]
        for e in range(epochs):
            for _ in range(spe):
                sched.step()
                rows.append([opt.param_groups[0]["lr"], opt.param_groups[1]["lr"], sched.current_lr])
            sched.step_epoch(bool(is_val[e]), bool(improved[e]))
            rows.append([opt.param_groups[0]["lr"], opt.param_groups[1]["lr"], sched.current_lr])
        out[name] = np.array(rows, dtype=np.float64)
        out[name + "_reductions"] = np.int64(getattr(sched, "reduce_steps", 0))
    import json
    out["cases_json"] = np.array(json.dumps([[n, sc, spe, ep] for n, sc, spe, ep in LR_SCHEDULE_CASES]))
    for ci, (name, _sc, _spe, epochs) in enumerate(LR_SCHEDULE_CASES):
        is_val, improved = lr_schedule_flags(ci, epochs)
        out[name + "_flags"] = np.stack([is_val, improved]).astype(np.int8)
    np.savez_compressed(os.path.join(OUT, "lr_schedule.npz"), **out)
    print("wrote lr_schedule")


def gen_collate():
    """RetrievalDataset.collate_fn (coot/dataset_retrieval.py:335-463; it never touches self) on seeded data points: every
    tensor of the batch it returns.  Two batches: ragged, and a single video with a single clip."""
    from coot.dataset_retrieval import RetrievalDataPointTuple, RetrievalDataset
    out = {}
    for name, (seed, B, dv, dt) in dict(ragged=(5, 6, 12, 10), single=(6, 1, 8, 4)).items():
        pts = []
        for d in O.make_datapoints(seed, B, dv, dt):
            par = th.from_numpy(d["par_feat"])
            sents, ptr = [], 0
            for n in d["sent_feat_len_list"]:
                sents.append(par[ptr:ptr + n]); ptr += n
            clips = [th.from_numpy(c) for c in d["clip_feat_list"]]
            pts.append(RetrievalDataPointTuple(d["key"], d["key"], ["w"] * len(clips), th.from_numpy(d["vid_feat"]),
                                               d["vid_feat"].shape[0], par, par.shape[0], len(clips), clips,
                                               [c.shape[0] for c in clips], len(clips), sents, d["sent_feat_len_list"]))
        batch = RetrievalDataset.collate_fn(None, pts)
        for k, v in batch.dict().items():
            if isinstance(v, th.Tensor):
                out[f"{name}_{k}"] = v.numpy()
        out[f"{name}_args"] = np.array([seed, B, dv, dt])
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **out)
    print("wrote collate")


def gen_mask_semantics():
    """Numeric version of tests_nntrainer/test_transformers.py:22-79: perturbing masked inputs
    must not change un-masked outputs of the encoder; we store outputs before/after."""
    cfg_o = O.NetConfig(input_dim=24, hidden_dim=32, num_heads=4, ff_dim=32, pool_hidden=64)
    P = O.make_params(cfg_o, 3)
    tc = dict(hidden_dim=32, num_layers=1, dropout=0.0, num_heads=4, pointwise_ff_dim=32, activation="gelu",
              norm="layernorm_coot")
    enc = models.transformer_legacy.TransformerEncoder(models.transformer_legacy.TransformerEncoderConfig(tc))
    sd = enc.state_dict()
    for k in sd:
        sd[k] = th.from_numpy(P["tf." + k].astype(np.float32))
    enc.load_state_dict(sd)
    enc.eval()
    rs = np.random.RandomState(9)
    x = rs.randn(3, 6, 32).astype(np.float32)
    lens = np.array([6, 4, 2])
    mask = th.from_numpy(np.arange(6)[None, :] >= lens[:, None])
    y0 = enc(th.from_numpy(x), mask).detach().numpy()
    x2 = x.copy()
    x2[np.arange(6)[None, :] >= lens[:, None]] += 3.0
    y1 = enc(th.from_numpy(x2), mask).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "mask_semantics.npz"), x=x, x2=x2, lens=lens, y0=y0, y1=y1)
    print("wrote mask_semantics")


ANET_DIMS = (2048, 1536, 384, 8, 384, 768)


anet_like_counts = O.anet_like_counts


def gen_full_small():
    """full path, small dims, full gradients"""
    gen_full("full_small", (40, 24, 32, 4, 32, 64), B=4, counts=[2, 1, 3, 2], Ls=(9, 7, 8, 5), seed=21, full_grads=True)


def gen_full_anet():
    """full path, ActivityNet paper dims (d_model 384, 8 heads, Dv 2048, Dt 1536), sub-sampled grads"""
    gen_full("full_anet", (2048, 1536, 384, 8, 384, 768), B=6, counts=[3, 1, 4, 2, 2, 5], Ls=(20, 16, 18, 9), seed=31,
             full_grads=False)


def gen_bench_anet():
    """BASELINE.json configs[1] exactly as bench.py runs it: 64 videos x 4 clips, Lv = Lc = 80, Lp = 64, Ls = 16 (fixed shape)."""
    gen_full("bench_anet", ANET_DIMS, B=64, counts=[4] * 64, Ls=(80, 80, 64, 16), seed=41, full_grads=False, ragged=False,
             sub_step=197, store_reshape=False)


def gen_bench_anet_ragged():
    """Same dims, ragged: clip counts ~ ANet annotation statistics, frame / word counts uniform up to the maxima."""
    gen_full("bench_anet_ragged", ANET_DIMS, B=64, counts=anet_like_counts(43, 64), Ls=(80, 80, 64, 30), seed=43,
             full_grads=False, ragged=True, sub_step=197, store_reshape=False)


def gen_bench_yc2_100m():
    """synthetic.WORKLOADS['yc2_100m'] (yc2_100m_coot.yaml: Dv 512, 16 videos x 8 clips, cycle weight 0.001)."""
    gen_full("bench_yc2_100m", (512, 1536, 384, 8, 384, 768), B=16, counts=[8] * 16, Ls=(80, 20, 96, 12), seed=47,
             full_grads=False, ragged=False, cc_weight=0.001, sub_step=197, store_reshape=False)


def gen_bench_anet_train():
    """bench_anet in TRAIN mode — the mode bench.py times — at dropout 0.1 on every site (the shipped 0.025 would hide a
    misplaced site under the bf16 noise), masks = the library's for step seed 20250926."""
    gen_full("bench_anet_train", ANET_DIMS, B=64, counts=[4] * 64, Ls=(80, 80, 64, 16), seed=41, full_grads=False, ragged=False,
             sub_step=197, store_reshape=False, train=dict(p=0.1, step_seed=20250926))


def gen_bench_anet_ragged_train():
    """The ragged batch in TRAIN mode, padded token rows (the reference's layout: padded positions draw masks too)."""
    gen_full("bench_anet_ragged_train", ANET_DIMS, B=64, counts=anet_like_counts(43, 64), Ls=(80, 80, 64, 30), seed=43,
             full_grads=False, ragged=True, sub_step=197, store_reshape=False, train=dict(p=0.1, step_seed=77001))


def gen_bench_anet_ragged_train_packed():
    """... and with the masks of the PACKED token-row layout (cu_seqlens: what bench.py --workload anet_ragged runs)."""
    gen_full("bench_anet_ragged_train_packed", ANET_DIMS, B=64, counts=anet_like_counts(43, 64), Ls=(80, 80, 64, 30), seed=43,
             full_grads=False, ragged=True, sub_step=197, store_reshape=False, train=dict(p=0.1, step_seed=77002, packed=True))


def gen_bench_hbm_stress():
    """BASELINE.json configs[4], per-GPU slice as synthetic.WORKLOADS['hbm_stress']: 16 videos x 64 clips x 80 frames x d = 1024
    (Cmax = 64: the global networks on 64-row sequences)."""
    gen_full("bench_hbm_stress", (1024, 1536, 384, 8, 384, 768), B=16, counts=[64] * 16, Ls=(80, 80, 64, 16), seed=59,
             full_grads=False, ragged=False, sub_step=197, store_reshape=False)


def gen_bench_hbm_stress_train():
    """bench_hbm_stress in TRAIN mode: 64 clips per video put the global networks on their per-op kernels (the single-launch passes
    take <= 32 items per sequence), so this pins the dropout sites of THAT path (GEMM epilogues, LayerNorm, attention kernels)."""
    gen_full("bench_hbm_stress_train", (1024, 1536, 384, 8, 384, 768), B=16, counts=[64] * 16, Ls=(80, 80, 64, 16), seed=59,
             full_grads=False, ragged=False, sub_step=197, store_reshape=False, train=dict(p=0.1, step_seed=424242))


def gen_bench_yc2_100m_2layer_train():
    """The 2-layer local encoders in TRAIN mode: the second layer's sites (site base 16) on the fused chains."""
    gen_full("bench_yc2_100m_2layer_train", (512, 1536, 384, 8, 384, 768), B=16, counts=[8] * 16, Ls=(80, 20, 96, 12), seed=61,
             full_grads=False, ragged=False, cc_weight=0.001, sub_step=197, store_reshape=False, layers=2,
             train=dict(p=0.1, step_seed=31337))


def gen_bench_yc2_100m_2layer():
    """BASELINE.json configs[0] as it words it: YouCook2-100m, batch 16, 2-layer local / 1-layer global encoders (d_model 384)."""
    gen_full("bench_yc2_100m_2layer", (512, 1536, 384, 8, 384, 768), B=16, counts=[8] * 16, Ls=(80, 20, 96, 12), seed=61,
             full_grads=False, ragged=False, cc_weight=0.001, sub_step=197, store_reshape=False, layers=2)


def gen_bench_yc2_2d3d():
    """synthetic.WORKLOADS['yc2_2d3d'] (yc2_2d3d_coot.yaml: Dv 4096, 64 videos x 8 clips)."""
    gen_full("bench_yc2_2d3d", (4096, 1536, 384, 8, 384, 768), B=64, counts=[8] * 64, Ls=(80, 20, 96, 12), seed=53,
             full_grads=False, ragged=False, cc_weight=0.001, sub_step=197, store_reshape=False)


def gen_bench_yc2_2d3d_2816():
    """BASELINE.json configs[3] AS IT WORDS IT: the concatenated 2D + 3D YouCook2 features at d = 2816 (the shipped YAML says 4096;
    synthetic.WORKLOADS['yc2_2d3d_2816'], `bench.py --workload yc2_2d3d_2816`), 64 videos x 8 clips.  TRAIN mode with the library's masks
    (p = 0.1): the fixture pins the timed configuration — dropout on — and with it the K = 2816 input FC (44 slabs of 64 columns:
    not a multiple of the 128-column double slab) and the 11-chunk-per-lane input LayerNorm."""
    gen_full("bench_yc2_2d3d_2816_train", (2816, 1536, 384, 8, 384, 768), B=64, counts=[8] * 64, Ls=(80, 20, 96, 12), seed=67,
             full_grads=False, ragged=False, cc_weight=0.001, sub_step=197, store_reshape=False, train=dict(p=0.1, step_seed=2816001))


RK_DIMS = (2048, 1536, 384, 8, 384, 768)
RK_N, RK_BATCH, RK_LS = 1024, 64, (40, 40, 32, 12)   # validation videos, batch, (Lv, Lc, Lp, Ls)
RK_EVAL = (4.0, 32, 0.3)    # validation set: feature noise, latent clusters, individual spread (make_latent_batch) — hard enough
                            # that R@1 is well below 1 (near-ties inside a cluster), far above chance


def rk_batch(seed, B=RK_BATCH):
    """One seeded batch of the R@K parity set: ANet-like clip counts, ragged lengths, every clip / sentence pair generated from a
    shared latent code (O.make_latent_batch), so that a briefly trained model retrieves far above chance."""
    Lv, Lc, Lp, Ls = RK_LS
    return O.make_latent_batch(seed, B, anet_like_counts(seed + 1, B), Lv, Lc, Lp, Ls, RK_DIMS[0], RK_DIMS[1])


def quantize_state(sd):
    """int8 per-tensor quantisation of a state dict (the fixture must stay small): returns (q, scale) per tensor and the
    de-quantised fp32 state BOTH implementations load."""
    q, deq = {}, {}
    for k, v in sd.items():
        a = v.detach().numpy().astype(np.float32)
        if k.endswith("embedding.pe"):
            continue
        sc = float(np.abs(a).max()) / 127.0 or 1.0
        qi = np.clip(np.round(a / sc), -127, 127).astype(np.int8)
        q[k] = (qi, np.float32(sc))
        deq[k] = qi.astype(np.float32) * np.float32(sc)
    return q, deq


def rk_eval_batch(i, B=RK_BATCH):
    Lv, Lc, Lp, Ls = RK_LS
    seed = 900000 + 13 * i
    return O.make_latent_batch(seed, B, anet_like_counts(seed + 1, B), Lv, Lc, Lp, Ls, RK_DIMS[0], RK_DIMS[1], noise=RK_EVAL[0],
                               clusters=RK_EVAL[1], spread=RK_EVAL[2])


def gen_rk_parity(train_steps=160):
    """SURVEY 8d retrieval-parity set: 1 024 videos (~3 800 clips), correlated features, a 'trained-like' state = the reference
    trained for `train_steps` Adam steps on seeded batches of the same distribution (its own modules, losses and optimizer
    settings, dropout on), then int8-quantised so the fixture stays small.  Stored: the quantised state, and R@1/5/10/50,
    MedR, MeanR of the REFERENCE's eval embeddings with that state (both directions, both levels) through
    nntrainer/retrieval.py, plus a few embedding rows."""
    cfg = ref_config(*RK_DIMS)
    ocfgs = oracle_cfgs(*RK_DIMS)
    th.manual_seed(0)
    mgr = model_retrieval.RetrievalModelManager(cfg)     # reference init (truncnorm 0.01)
    tr = _FakeTrainer(cfg)
    params = [p for k in NET_KEYS for p in mgr.model_dict[k].parameters() if p.requires_grad]
    opt = th.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=2e-5)
    mgr.set_all_models_train()
    import time
    t0 = time.time()
    prev = os.path.join(OUT, "rk_parity.npz")
    reuse = os.path.exists(prev) and not os.environ.get("RK_RETRAIN")
    if reuse:  # the trained state of the committed fixture (13 CPU-minutes to reproduce: RK_RETRAIN=1): only the evaluation is redone
        gp = np.load(prev)
        for k in NET_KEYS:
            sd = mgr.model_dict[k].state_dict()
            for n in list(sd):
                if f"q:{k}:{n}" in gp.files:
                    sd[n] = th.from_numpy(gp[f"q:{k}:{n}"].astype(np.float32) * np.float32(gp[f"s:{k}:{n}"]))
            mgr.model_dict[k].load_state_dict(sd)
        train_steps = int(gp["meta"][-1])
        print("  rk_parity: re-using the trained state of the existing fixture")
    for step in range(0 if reuse else train_steps):
        batch = to_batch(rk_batch(5000 + 7 * step))
        opt.zero_grad()
        vis, txt = mgr.encode_visual(batch), mgr.encode_text(batch)
        loss = tr.compute_total_constrastive_loss(vis, txt) + tr.compute_cyclecons_loss(vis, txt)
        loss.backward()
        opt.step()
        if step % 20 == 0:
            print(f"  rk_parity train step {step}: loss {float(loss):.4f} ({time.time() - t0:.0f} s)", flush=True)
    out = {}
    for k in NET_KEYS:
        q, deq = quantize_state(mgr.model_dict[k].state_dict())
        sd = mgr.model_dict[k].state_dict()
        for n, v in deq.items():
            sd[n] = th.from_numpy(v)
        mgr.model_dict[k].load_state_dict(sd)
        for n, (qi, sc) in q.items():
            out[f"q:{k}:{n}"] = qi
            out[f"s:{k}:{n}"] = sc
    mgr.set_all_models_eval()
    embs = {k: [] for k in ("vid_emb", "par_emb", "clip_emb", "sent_emb")}
    with th.no_grad():
        for i in range(RK_N // RK_BATCH):
            batch = to_batch(rk_eval_batch(i))
            vis, txt = mgr.encode_visual(batch), mgr.encode_text(batch)
            for k, v in (("vid_emb", vis.vid_emb), ("par_emb", txt.par_emb), ("clip_emb", vis.clip_emb), ("sent_emb", txt.sent_emb)):
                embs[k].append(v.numpy())
    E = {k: np.concatenate(v) for k, v in embs.items()}
    for (a, c, tag) in (("vid_emb", "par_emb", "vp"), ("clip_emb", "sent_emb", "cs")):
        e1 = E[a] / np.sqrt((E[a] ** 2).sum(-1, keepdims=True))      # coot/trainer_retrieval.py:401-402 (no eps)
        e2 = E[c] / np.sqrt((E[c] ** 2).sum(-1, keepdims=True))
        r1, r2, s1 = retrieval.compute_retrieval({"a": e1, "b": e2}, "a", "b", print_fn=lambda *_: None)[:3]
        keys = ("r1", "r5", "r10", "r50", "medr", "meanr")
        out[f"ret_{tag}"] = np.array([r1[k] for k in keys] + [r2[k] for k in keys] + [s1])
        print(f"  rk_parity {tag}: N = {len(e1)}  " + "  ".join(f"{k} {r1[k]:.4f}/{r2[k]:.4f}" for k in keys))
    out["n_clips"] = np.array(len(E["clip_emb"]))
    for k in E:
        out["rows:" + k] = E[k][::37].astype(np.float32)
    out["meta"] = np.array([RK_N, RK_BATCH, *RK_LS, *RK_DIMS, train_steps])
    out["eval_gen"] = np.array(RK_EVAL, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "rk_parity.npz"), **out)
    print("wrote rk_parity")


# ---- a composed training trajectory: k optimizer steps of the reference's own step body ------------------------------------
from nntrainer import optimization  # noqa: E402


def gen_train_trajectory(name, dims, B, counts, Ls, seed, steps, p, step_seed0, scale=0.05, ragged=True, cc_weight=None, adam_eps=None,
                         full=True, sub_step=29, layers=1, packed=False, optimizer=None):
    """`steps` consecutive optimizer steps of the reference's step body (coot/trainer_retrieval.py:253-291: zero_grad, encode_visual,
    encode_text, total contrastive + cycle-consistency loss, backward, optimizer.step) on TWO seeded batches used in turn (as bench.py
    does: the loss of step s then shows what steps s - 2, s - 4, ... did to the parameters), with the
    optimizer the reference builds for the shipped configuration (nntrainer/optimization.py:45-74 over the parameter groups of
    nntrainer/models/model_manager_base.py:130-164: Adam lr 1e-3, betas (0.9, 0.999), weight decay 2e-5 with decay_mult 0 on every
    parameter whose name contains 'bias' — weight_decay_for_bias: true, sic; no gradient clipping: anet_coot.yaml clip_gradient -1).
    TRAIN mode at dropout p, the masks being the library's for the step seeds step_seed0 + 7919 s (oracle/dropout_masks.py), the
    cycle-consistency positions drawn by th.multinomial under th.manual_seed(seed + 7 + s) and stored.
    Stored: the two loss values of every step and final - initial of every parameter (full, or sub-sampled + norms).
    adam_eps: the shipped 1e-8 makes the first updates +-lr whatever the gradient's size (a sign function: it amplifies any
    difference between two implementations' small gradient entries); the fixtures also come at eps = 1e-3 (>= the typical
    gradient entry), where the update is a smooth function of the gradient and a per-tensor cosine bound has power."""
    dv, dt, hidden, heads, ff, pool_hidden = dims
    Lv, Lc, Lp, Lsent = Ls
    cfg = ref_config(*dims, layers=layers, dropout=p)
    if cc_weight is not None:
        cfg.train.loss_cycle_cons = cc_weight
    for k_, v_ in (optimizer or {}).items():  # e.g. the YouCook2 configurations' RAdam (yc2_100m_coot.yaml:136-148)
        assert hasattr(cfg.optimizer, k_), k_
        setattr(cfg.optimizer, k_, v_)
    if adam_eps is not None:
        cfg.optimizer.adam_eps = adam_eps
    assert cfg.optimizer.name in ("adam", "radam") and cfg.train.clip_gradient == -1
    ocfgs = oracle_cfgs(*dims, layers=layers)
    th.manual_seed(0)
    mgr = model_retrieval.RetrievalModelManager(cfg)
    for i, k in enumerate(NET_KEYS):
        load_params(mgr.model_dict[k], O.make_params(ocfgs[i], seed + 10 * i, scale=scale))
    mgr.set_all_models_train()
    # packed: the masks of the PACKED token-row layout of the local networks (cu_seqlens: what bench.py --workload anet_ragged runs)
    states = [inject_dropout(mgr.model_dict[k], 0, float(p), packed=packed and k.endswith("local")) for k in NET_KEYS]
    params, _names, _flat = mgr.get_all_params()
    opt = optimization.make_optimizer(cfg.optimizer, params)
    init = {(k, n): q.detach().clone() for k in NET_KEYS for n, q in mgr.model_dict[k].named_parameters()}
    tr = _FakeTrainer(cfg)
    losses, idxs, seeds = [], [], []
    for s in range(steps):
        step_seed = int(step_seed0) + 7919 * s
        for stt, net_seed in zip(states, DM.step_net_seeds(step_seed)):
            stt.update(seed=net_seed, calls=0, row_next=0, tok_next=0, layout=None)
        batch = to_batch(O.make_batch(seed + 100 + (s & 1), B, counts, Lv, Lc, Lp, Lsent, dv, dt, ragged=ragged, corr=0.5))
        opt.zero_grad()
        vis = mgr.encode_visual(batch)
        txt = mgr.encode_text(batch)
        contr = tr.compute_total_constrastive_loss(vis, txt)
        ic, isent = draw_cc_indices(seed + 7 + s, vis.clip_emb_mask, txt.sent_emb_mask)
        th.manual_seed(seed + 7 + s)
        cc = tr.compute_cyclecons_loss(vis, txt)
        (contr + cc).backward()
        opt.step()
        losses.append([float(contr), float(cc)])
        idxs.append(np.stack([ic, isent]))
        seeds.append(step_seed)
        print(f"  {name} step {s}: contrastive {float(contr):.5f} cycle-consistency {float(cc):.6f}", flush=True)
    out = dict(losses=np.array(losses, dtype=np.float64), cc_idx=np.array(idxs, dtype=np.int64), step_seeds=np.array(seeds, dtype=np.uint64),
               meta=np.array([seed, B, Lv, Lc, Lp, Lsent, dv, dt, hidden, heads, ff, pool_hidden]), ragged=np.array(int(ragged)),
               cc_weight=np.array(float(cfg.train.loss_cycle_cons)), param_scale=np.array(scale), counts=np.asarray(counts),
               layers=np.array(layers), train_p=np.array(float(p)), steps=np.array(steps), sub_step=np.array(sub_step),
               train_packed=np.array(int(bool(packed))), opt_name=np.array(cfg.optimizer.name),
               radam_degentosgd=np.array(int(bool(cfg.optimizer.radam_degentosgd))),
               adam=np.array([cfg.optimizer.lr, cfg.optimizer.momentum, cfg.optimizer.adam_beta2, cfg.optimizer.adam_eps,
                              cfg.optimizer.weight_decay, float(cfg.optimizer.weight_decay_for_bias)], dtype=np.float64))
    for k in NET_KEYS:
        for n, q in mgr.model_dict[k].named_parameters():
            d = (q.detach() - init[(k, n)]).numpy()
            out[f"dnorm:{k}:{n}"] = np.array(np.linalg.norm(d.astype(np.float64)))
            out[f"delta:{k}:{n}"] = d if (full or d.size <= 4096) else subsample(d, sub_step)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name)


TRAJ_SMALL = dict(dims=(64, 48, 64, 4, 64, 128), B=4, counts=[2, 1, 3, 2], Ls=(9, 7, 8, 5), seed=71, steps=8, p=0.1, step_seed0=5150001)  # (the library's d_head: 16, 32, 48, 64)
TRAJ_ANET = dict(dims=ANET_DIMS, B=64, counts=[4] * 64, Ls=(80, 80, 64, 16), seed=73, steps=3, p=0.1, step_seed0=5150101, ragged=False,
                 full=False)


def gen_traj_small():
    """8 optimizer steps at small dims (the per-op kernels), every parameter's final - initial in full, shipped Adam (eps 1e-8)."""
    gen_train_trajectory("traj_small", **TRAJ_SMALL)


def gen_traj_small_eps():
    """... and at eps = 1e-3 (smooth updates: the bound on the parameter deltas is tight there)."""
    gen_train_trajectory("traj_small_eps", adam_eps=1e-3, **TRAJ_SMALL)


YC2_RADAM = dict(name="radam", lr=9e-4, weight_decay=0.0, momentum=0.56, adam_beta2=0.98, adam_eps=1.5e-9, radam_degentosgd=False)


def gen_traj_small_radam():
    """10 optimizer steps with the YouCook2 configurations' optimizer (yc2_100m_coot.yaml / yc2_2d3d_coot.yaml:136-148: the in-file RAdam of
    nntrainer/optimization.py:79-181, lr 9e-4, betas (0.56, 0.98), eps 1.5e-9, no weight decay, not degenerated to SGD): with beta2 =
    0.98 the variance rectification switches on at step 6 — steps 1-5 move only the moments, steps 6-10 the parameters."""
    gen_train_trajectory("traj_small_radam", optimizer=YC2_RADAM, **dict(TRAJ_SMALL, steps=10, seed=83, step_seed0=5150301))


def gen_traj_small_radam_eps():
    gen_train_trajectory("traj_small_radam_eps", optimizer=dict(YC2_RADAM, adam_eps=1e-3), **dict(TRAJ_SMALL, steps=10, seed=83, step_seed0=5150301))


def gen_traj_yc2_100m_radam_eps():
    """BASELINE.json configs[0] as it words it (YouCook2-100m shapes, 2-layer local / 1-layer global encoders, batch 16 x 8 clips,
    d_model 384: the fused chains, cycle weight 0.001) trained for 8 steps with that configuration's RAdam — steps 6-8 move the
    parameters — at eps = 1e-3."""
    gen_train_trajectory("traj_yc2_100m_radam_eps", optimizer=dict(YC2_RADAM, adam_eps=1e-3), dims=(512, 1536, 384, 8, 384, 768), B=16, counts=[8] * 16,
                         Ls=(80, 20, 96, 12), seed=89, steps=8, p=0.1, step_seed0=5150401, ragged=False, cc_weight=0.001, full=False, layers=2)


def gen_traj_anet():
    """3 optimizer steps at the benchmark's shapes (64 videos x 4 clips, d_model 384: the fused chains), shipped Adam."""
    gen_train_trajectory("traj_anet", **TRAJ_ANET)


def gen_traj_anet_eps():
    gen_train_trajectory("traj_anet_eps", adam_eps=1e-3, **TRAJ_ANET)


def gen_traj_anet_ragged_packed_eps():
    """3 optimizer steps on RAGGED ActivityNet-shaped batches (clip counts ~ annotation statistics, ragged lengths) with the masks of the
    packed token-row layout — the composed loop of `bench.py --workload anet_ragged` — at eps = 1e-3."""
    gen_train_trajectory("traj_anet_ragged_packed_eps", adam_eps=1e-3, dims=ANET_DIMS, B=64, counts=anet_like_counts(43, 64), Ls=(80, 80, 64, 30), seed=79,
                         steps=3, p=0.1, step_seed0=5150201, ragged=True, full=False, packed=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]  # e.g. "lr_schedule": regenerate just that fixture
    if only:
        for name in only:
            globals()["gen_" + name]()
        return
    small = O.NetConfig(input_dim=40, hidden_dim=32, num_heads=4, ff_dim=32, pool_hidden=64)
    gen_single_net("net_local_small", small, N=5, L=7, seed=11, with_ctx=False)
    smallg = O.NetConfig(input_dim=32, hidden_dim=32, num_heads=4, ff_dim=32, pool_hidden=64,
                         use_input_fc=False, use_context=True, pooler="avg_special")
    gen_single_net("net_global_small", smallg, N=4, L=5, seed=13, with_ctx=True)
    two = O.NetConfig(input_dim=40, hidden_dim=32, num_heads=4, ff_dim=48, pool_hidden=32, num_layers=2)
    gen_single_net("net_local_2layer", two, N=3, L=6, seed=17, with_ctx=False)
    gen_full_small()
    gen_full_anet()
    gen_bench_anet()
    gen_bench_anet_ragged()
    gen_bench_yc2_100m()
    gen_bench_yc2_2d3d()
    gen_bench_anet_train()
    gen_bench_anet_ragged_train()
    gen_bench_anet_ragged_train_packed()
    gen_bench_hbm_stress()
    gen_bench_yc2_100m_2layer()
    gen_bench_hbm_stress_train()
    gen_bench_yc2_100m_2layer_train()
    gen_bench_yc2_2d3d_2816()
    gen_traj_small()
    gen_traj_small_eps()
    gen_traj_anet()
    gen_traj_anet_eps()
    gen_traj_anet_ragged_packed_eps()
    gen_traj_small_radam()
    gen_traj_small_radam_eps()
    gen_traj_yc2_100m_radam_eps()
    gen_rk_parity()
    gen_retrieval_metrics()
    gen_radam()
    gen_lr_schedule()
    gen_collate()
    gen_mask_semantics()


if __name__ == "__main__":
    main()
