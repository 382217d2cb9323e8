"""`bench.py --gpus 8` end to end with EIGHT real processes on the one GPU of the test box (test mode: --share-device, gloo with
host-staged collectives — RCCL refuses two ranks on one device).  What an 8-GPU run of the driver executes, minus RCCL itself:
the launcher (`python -m torch.distributed.run ... bench.py --gpus 8`, and bench.py's own spawn_ranks), the 127.0.0.1 rendezvous,
one trainer + library state per process, the embedding-block all-gather with R = 8 block offsets, the contrastive loss on the
gathered batch (this rank's strips against 512 videos / 2 048 clips), the three gradient buckets, the update, rank 0's one JSON line.
nntrainer/trainer_base.py:126-129 semantics: encoders per shard, loss on the full batch — so the job's contrastive loss must equal
a single-process native step on the UNION batch (no dropout, learning rate 0: --no-dropout --lr0).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _union_contrastive(world):
    """The contrastive loss of ONE process on the union of the ranks' batches (bench.py: rank r's batch is make_batch(1234 + r, ...))."""
    import torch
    import coot_videotext_amd as cva
    w = cva.synthetic.WORKLOADS["anet"]
    cfg = cva.load_named_config(*cva.synthetic.WORKLOAD_CONFIG["anet"])
    cfg.optimizer.lr, cfg.optimizer.weight_decay = 0.0, 0.0
    torch.manual_seed(0)
    mgr = cva.RetrievalModelManager(cfg).cuda()
    trainer = cva.RetrievalTrainer(cfg, mgr)
    mgr.set_all_models_eval()
    parts = [cva.synthetic.make_batch(1234 + r, w["B"], w["C"], w["Lv"], w["Lc"], w["Lp"], w["Ls"], w["Dv"], w["Dt"], ragged=False) for r in range(world)]
    cat = lambda f: torch.cat([getattr(p, f) for p in parts])
    B = w["B"] * world
    keys = [str(i) for i in range(B)]
    union = cva.RetrievalDataBatchTuple(
        key=keys, data_key=keys, sentences=[[""]] * B, vid_feat=cat("vid_feat"), vid_feat_mask=cat("vid_feat_mask"), vid_feat_len=cat("vid_feat_len"),
        par_feat=cat("par_feat"), par_feat_mask=cat("par_feat_mask"), par_feat_len=cat("par_feat_len"), clip_num=cat("clip_num"),
        clip_feat=cat("clip_feat"), clip_feat_mask=cat("clip_feat_mask"), clip_feat_len=cat("clip_feat_len"), sent_num=cat("sent_num"),
        sent_feat=cat("sent_feat"), sent_feat_mask=cat("sent_feat_mask"), sent_feat_len=cat("sent_feat_len"), max_clip_num=w["C"], max_sent_num=w["C"])
    del parts
    losses = trainer.train_step_native(union, do_optimizer=False, seed=1)
    torch.cuda.synchronize()
    return float(losses[1])


def _check_line(out, world):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]  # rank 0's stdout carries exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["config"]["parallelism"] == f"dp{world}" and d["config"]["share_device"] is True
    assert d["config"]["clip_pairs_per_step"] == 256 * world and d["config"]["global_batch_videos"] == 64 * world
    assert d["value"] > 0 and d["scaling"] == "weak" and d["steps"] == 2
    return d


@pytest.mark.timeout(900)
def test_bench_gpus_8_as_the_driver_launches_it():
    world = 8
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--dp-backend", "gloo", "--share-device", "--no-dropout", "--lr0", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _check_line(r.stdout, world)
    total, contr, cc = d["config"]["final_losses"]
    want = _union_contrastive(world)
    print(f"8 ranks: contrastive {contr:.6f} (union batch in one process {want:.6f}), cycle-consistency {cc:.6f}, total {total:.6f}")
    assert abs(contr - want) <= 2e-4 * abs(want), (contr, want)
    assert abs(total - contr - cc) < 1e-5


@pytest.mark.timeout(600)
def test_bench_gpus_2_self_spawned():
    """`python bench.py --gpus 2` without a launcher: bench.py starts the ranks itself (spawn_ranks)."""
    world = 2
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--dp-backend", "gloo",
           "--share-device", "--no-dropout", "--lr0", "--no-cpu-baseline", "--no-roofline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _check_line(r.stdout, world)
    contr = d["config"]["final_losses"][1]
    want = _union_contrastive(world)
    assert abs(contr - want) <= 2e-4 * abs(want), (contr, want)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("route", ["direct", "torch"])
def test_bench_one_rank_over_rccl_prints_one_json_line(route):
    """`bench.py --force-dp` with the real backend (nccl = RCCL, one rank: all this box can hold): the step's collectives as direct
    RCCL calls on its own streams (dist.DirectRccl; default) or through torch.distributed (COOT_DP_COLLECTIVES=torch).  STDOUT must
    be exactly the one JSON line — RCCL prints a version banner to stdout when a process creates its first communicator, so both
    communicators have to be created inside bench.py's redirection — and the line names the route that ran."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dp", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["COOT_DP_COLLECTIVES"] = route
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=500, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1" and d["value"] > 0
    assert d["config"]["collectives"].startswith("RCCL calls" if route == "direct" else "torch.distributed"), d["config"]["collectives"]
