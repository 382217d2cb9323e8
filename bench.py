#!/usr/bin/env python
"""
bench.py — COOT retrieval TRAINING step throughput on MI355X (clip-pairs/sec, whole job).

One step = encode_visual + encode_text + contrastive + cycle-consistency losses + backward (+ gradient
all-reduce for N > 1) + Adam step on one synthetic ActivityNet-shaped batch per GPU (BASELINE.json configs[1]:
B = 64 videos x 4 clips/GPU, Lc = Lv = 80 frames, Ls = 16 / Lp = 64 tokens, Dv = 2048, Dt = 1536, d_model 384;
bf16 MFMA operands, fp32 accumulation/statistics/master weights; dropout ON as in training).  Inputs are
resident in HBM before the timed region.  Weak scaling: per-GPU batch fixed, global batch = 64 N videos.

    python bench.py [--gpus N --steps K --warmup W] [--workload anet|yc2_100m|yc2_2d3d|yc2_2d3d_2816|hbm_stress]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline": the MFMA kernel family the step spends most time in (the fused token-tile chains: infc_qkv_fwd, post_attn_fwd,
              pre_attn_bwd, qkv_bwd), algorithmic FLOP/s from HIP events on the launch stream vs 2.5 PF dense bf16;
              for --workload hbm_stress the input LayerNorm (the kernel that reads the feature stream) vs 8 TB/s HBM;
  "cpu_baseline": oracle/coot_torch_cpu.py (a PyTorch-CPU restatement issuing the reference's ATen ops: kind "port")
              timed on the host cores on the full per-GPU batch of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="anet")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="library A/B switch name=value (coot_set_option), repeatable")
    ap.add_argument("--force-dp", action="store_true", help="run the data-parallel step (phase calls + RCCL collectives) even with one rank")
    ap.add_argument("--dp-backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend of the data-parallel step: nccl = RCCL over xGMI "
                    "(the product path); gloo stages the collectives through host memory (dist.py: _host_staged) — for --share-device")
    ap.add_argument("--share-device", action="store_true", help="TEST MODE, not a benchmark: all N ranks run on device 0 (RCCL refuses two ranks on one "
                    "device, hence --dp-backend gloo).  Exercises the launcher, the rendezvous, the gloo side group, R-rank block offsets, the "
                    "three-bucket reduce and the JSON line exactly as an N-GPU run does, on a one-GPU box; the line carries share_device = true")
    ap.add_argument("--no-dropout", action="store_true", help="run the train step with the networks in eval mode (no dropout masks): the loss of a "
                    "data-parallel job is then comparable with a single-process step on the union batch (tests)")
    ap.add_argument("--step-stamps", action="store_true", help="print a HIP-event timeline of one training step to stderr")
    ap.add_argument("--clock-monitor", action="store_true", help="after the timed loop: sample the shader clock on a side stream while more steps run (stderr)")
    ap.add_argument("--no-lookahead", action="store_true", help="do not announce the next batch to the step: its input LayerNorm runs at the head of its own step instead of next to the previous step's global networks (A/B)")
    ap.add_argument("--no-defer-join", action="store_true", help="join the text stream into the main stream at the end of every step (A/B)")
    ap.add_argument("--repeat", type=int, default=1, help="regime probe: time R blocks of K steps back to back (each bracketed like the first); the JSON "
                    "line is the FIRST block (the contract's), the others go to stderr and to 'blocks_ms_per_step'")
    ap.add_argument("--per-step-events", action="store_true", help="regime probe: a timing event on the main stream behind every step (warm-up "
                    "included); per-step device times to stderr.  Perturbs the step slightly: not for reported numbers")
    ap.add_argument("--clock-monitor-early", action="store_true", help="regime probe: sample the shader clock (one wave on a side stream, every "
                    "100 us) from BEFORE the warm-up through the timed steps; to stderr")
    ap.add_argument("--pre-burn", default="", help="regime probe: KIND:MS — keep the device busy for MS milliseconds right before the warm-up with "
                    "'hbm' (1 GB device copies) or 'alu' (L2-resident elementwise launches); tests whether the slow first steps are a clock governor")
    ap.add_argument("--sysfs-clocks", action="store_true", help="regime probe: print the DPM clock tables of the device (sysfs) before the warm-up, after it "
                    "and after the timed steps, to stderr")
    ap.add_argument("--idle-ms", type=float, default=0.0, help="regime probe: leave the device idle for this long between the warm-up and the timed steps")
    ap.add_argument("--lr0", action="store_true", help="regime probe: learning rate and weight decay 0 — every step computes on the same weights")
    ap.add_argument("--extra-streams", type=int, default=0, help="regime probe: create and use N unrelated HIP streams before the trainer exists — shifts "
                    "HIP's stream -> hardware-queue mapping (creation order, 4 queues); the step's streams are verified to run concurrently "
                    "(coot_stream_create_concurrent), so the result must not depend on N (profiles/r06_stream_queues.txt)")
    ap.add_argument("--eval", action="store_true", help="forward-only (eval mode) throughput instead of training")
    ap.add_argument("--padded", action="store_true", help="ragged workloads: run the reference's padded layout instead of packed (varlen) rows")
    ap.add_argument("--mode", default="native", choices=["native", "native-graph", "autograd"],
                    help="native: one C call per step (default); native-graph: that call captured once and replayed as a hipGraph; "
                         "autograd: torch autograd Functions (eager)")
    return ap.parse_args()


def algorithmic_flops_ragged(lens, D=384):
    """The same count from the real lengths of a ragged batch: tokens beyond a sequence's length are not useful work
    (len tokens, len^2 attention); lens = dict of numpy arrays vid / clip / par / sent lengths and clip counts per video."""
    def loc(din, l):
        l = l.astype(np.float64)
        return float((l * (2 * din * D + 18 * D * D) + 4 * l * l * D).sum())

    fwd = loc(lens["Dv"], lens["clip"]) + loc(lens["Dv"], lens["vid"]) + loc(lens["Dt"], lens["sent"]) + loc(lens["Dt"], lens["par"])
    Cn = lens["counts"].astype(np.float64)
    fwd += 2 * float((Cn * (12 * D * D + 4 * Cn * D) + Cn * 4 * D * D + 8 * D * D + 4 * Cn * D).sum())
    B, Nc = len(Cn), float(Cn.sum())
    fwd += 2 * (3 * B * B * 768 + 3 * Nc * Nc * D + B * B * D)
    return fwd, 3 * fwd


def algorithmic_flops_per_step(w, cfg):
    """forward GEMM MACs x 2 of the four networks + losses; train = 3 x forward."""
    D = 384

    def f_loc(din, L):
        return 2 * din * D + 18 * D * D + 4 * L * D

    B, Cn = w["B"], w["C"]
    Nc = B * Cn
    fwd = (Nc * w["Lc"] * f_loc(w["Dv"], w["Lc"]) + B * w["Lv"] * f_loc(w["Dv"], w["Lv"]) +
           Nc * w["Ls"] * f_loc(w["Dt"], w["Ls"]) + B * w["Lp"] * f_loc(w["Dt"], w["Lp"]))
    glob = B * (Cn * (12 * D * D + 4 * Cn * D) + Cn * 4 * D * D + 8 * D * D + 4 * Cn * D)
    fwd += 2 * glob
    # 7 similarity GEMMs (3 align + 4 cluster), 2*N^2*d each (trainer_retrieval.py:168-182)
    fwd += 2 * (3 * B * B * 768 + 3 * Nc * Nc * D + B * B * D)
    return fwd, 3 * fwd


def cpu_baseline(w, budget_s=22.0, budget_1t_s=10.0):
    """CPU baseline ("port"): oracle/coot_torch_cpu.py — the train step restated with the same PyTorch CPU ops the
    reference's modules issue (fp32, autograd backward, torch.optim.Adam), dropout on, on the FULL per-GPU batch of the
    workload (same shapes as the GPU step).  Median step time over ~budget_s seconds of CPU work (at least 2 steps after a
    warm-up step); clip-pairs/s = clips per step / median step time.  Intra-op threads are capped at 32: on a many-core host
    more threads are SLOWER for these op sizes ("cores" = threads actually used).  A second, ONE-thread figure (SURVEY 8d asks
    for N = 1 next to N = all cores) on a bounded sample — the first quarter of the batch's videos with their clips, ~budget_1t_s
    seconds — rides along as "value_1thread".  The reference itself (/root/reference) does not exist on the GPU box; SURVEY 8d
    quotes its own time in the build container (136 clip-pairs/s on 8 cores, ANet shape)."""
    from oracle import coot_oracle as O
    from oracle import coot_torch_cpu as T
    from tests import helpers as H
    Bs = w["B"]
    dims = (w["Dv"], w["Dt"], 384, 8, 384, 768)
    cfgs = H.full_cfgs(*dims)
    Ps = [T.to_torch_params(O.make_params(cfgs[i], 5 + 10 * i, dtype=np.float32)) for i in range(4)]
    opt = torch.optim.Adam([v for P in Ps for k, v in P.items() if v.requires_grad], lr=1e-3, weight_decay=2e-5)

    def timed(nvid, threads, budget, max_steps):
        torch.set_num_threads(threads)
        counts = w["C"] if w["C"] else O.anet_like_counts(4321, nvid)   # C = 0: the ragged workload (the CPU path pads, as the reference does)
        b = O.make_batch(1, nvid, counts, w["Lv"], w["Lc"], w["Lp"], w["Ls"], w["Dv"], w["Dt"], ragged=not w["C"], dtype=np.float32)
        idx = np.zeros(nvid, dtype=np.int64)

        def one():
            T.full_step(cfgs, Ps, b, idx, idx, H.ANET_W, 0.2, 0.01, p_drop=0.025, train=True)
            opt.step()

        t_all = time.perf_counter()
        one()  # warm-up
        times = []
        while len(times) < 2 or (time.perf_counter() - t_all < budget and len(times) < max_steps):
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
        return float(np.sum(b["clip_num"])), float(np.median(times)), len(times)

    threads = max(1, min(32, os.cpu_count() or 1))
    clips, dt, n = timed(Bs, threads, budget_s, 30)
    nv1 = max(2, Bs // 4)
    clips1, dt1, n1 = timed(nv1, 1, budget_1t_s, 4)
    return {"value": clips / dt, "unit": "clip-pairs/s", "cores": threads, "kind": "port",
            "sample": f"PyTorch-CPU fp32 restatement (oracle/coot_torch_cpu.py: same ATen ops as the reference modules, autograd, "
                      f"Adam, dropout on), the full per-GPU batch: {Bs} videos, {int(clips)} clips of the same shapes, median of "
                      f"{n} steps, {dt:.3f} s/step, host cpu_count {os.cpu_count()}",
            "value_1thread": clips1 / dt1, "sample_1thread": f"the same step with torch.set_num_threads(1) on the first {nv1} videos "
                      f"({int(clips1)} clips), median of {n1} steps, {dt1:.3f} s/step"}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, LOCAL_RANK = device
    index, rendezvous on 127.0.0.1 at a free port) — what `python -m torch.distributed.run --nproc-per-node N` would do.  Rank 0
    inherits stdout and prints the JSON line.  Fails loudly when fewer than N devices are visible."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    share = "--share-device" in sys.argv
    if have < (1 if share else n):
        raise RuntimeError(f"bench.py --gpus {n}: only {have} GPU(s) visible — refusing to report a {n}-GPU number from fewer devices")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    return next((rc for rc in rcs if rc), 0)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device and args.dp_backend != "gloo":
        raise RuntimeError("bench.py --share-device needs --dp-backend gloo (RCCL refuses two ranks on one device)")
    if world != args.gpus:
        raise RuntimeError(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback in the product path)")
    if torch.cuda.device_count() <= local_rank:
        raise RuntimeError(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    import coot_videotext_amd as cva
    from coot_videotext_amd import dist as cdist
    lib = cva.lib.load()
    for kv in args.opt:
        k, v = kv.split("=")
        cva.lib.check(lib.coot_set_option(k.encode(), int(v)), "coot_set_option")
    dp = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints its version banner to STDOUT when the first communicator is created: keep it off the JSON line's channel
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(args.dp_backend, rank=rank, world_size=world)
            dist.barrier()  # creates the communicator now, inside the redirection
            dp = cdist.DataParallelContext()
            dp.prepare_device_collectives()  # ... and the step's own (direct RCCL calls on its streams: dist.DirectRccl)
            torch.cuda.synchronize()
        finally:
            C.CDLL(None).fflush(None)  # RCCL writes through C stdio: its buffer must be drained while fd 1 still is stderr
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        assert dist.get_world_size() == world
    w = cva.synthetic.WORKLOADS[args.workload]
    cfg = cva.load_named_config(*cva.synthetic.WORKLOAD_CONFIG[args.workload])
    torch.manual_seed(0)  # identical initial weights on every rank
    if args.lr0:
        cfg.optimizer.lr, cfg.optimizer.weight_decay = 0.0, 0.0
    mgr = cva.RetrievalModelManager(cfg).cuda()
    _extra = [torch.cuda.Stream() for _ in range(args.extra_streams)]
    for _s in _extra:
        with torch.cuda.stream(_s):
            torch.zeros(1, device="cuda")
    trainer = cva.RetrievalTrainer(cfg, mgr, is_test=args.eval)
    if dp is not None:
        trainer.dp = dp
        trainer.comm_stream = torch.cuda.Stream()
    ragged = w["C"] == 0
    if ragged:  # every rank knows every rank's clip counts (seeded): no collective for the shard sizes or the global Cmax
        all_counts = [cva.synthetic.anet_like_counts(4321 + r, w["B"]) for r in range(world)]
        batch = cva.synthetic.make_batch(1234 + rank, w["B"], all_counts[rank], w["Lv"], w["Lc"], w["Lp"], w["Ls"], w["Dv"], w["Dt"], ragged=True,
                                         packed=not args.padded)
        batch.max_clip_num = batch.max_sent_num = int(max(c.max() for c in all_counts))
        clip_counts = [int(c.sum()) for c in all_counts]
    else:
        batch = cva.synthetic.make_batch(1234 + rank, w["B"], w["C"], w["Lv"], w["Lc"], w["Lp"], w["Ls"], w["Dv"], w["Dt"], ragged=False)
        clip_counts = [w["B"] * w["C"]] * world
    vid_counts = [w["B"]] * world
    clip_pairs = sum(clip_counts)

    la_on = [False]
    if args.eval:
        mgr.set_all_models_eval()

        def step():
            with torch.no_grad():
                v = mgr.encode_visual(batch)
                t = mgr.encode_text(batch)
            return v.vid_emb.sum() + t.par_emb.sum()
    else:
        mgr.set_all_models_train()
        if args.no_dropout:
            mgr.set_all_models_eval()  # train_step_native passes train = model_mgr.is_train: same step, no masks

        mode = args.mode
        batch.global_max_synced = True  # fixed shapes: every rank has the same Cmax, no MAX all-reduce needed
        two_batches = mode == "native"
        # (also on the data-parallel phase path: the next batch's input LayerNorm runs under the embedding exchange and the loss)
        lookahead = two_batches and not args.no_lookahead
        if two_batches:
            # Two DIFFERENT synthetic batches of the workload's shape, used in turn (with and without the lookahead): a step never sees the
            # data of the step before it (a single resident batch would partly live in the 256 MB MALL from step to step), and the batch a
            # step announces as "next" — whose input LayerNorm it runs next to its global networks — is other data than the one it trains
            # on: nothing a step computes for itself is ever reused, every step executes one input LayerNorm per side
            if ragged:
                other = cva.synthetic.make_batch(9234 + rank, w["B"], all_counts[rank], w["Lv"], w["Lc"], w["Lp"], w["Ls"], w["Dv"], w["Dt"], ragged=True,
                                                 packed=not args.padded)
                other.max_clip_num = other.max_sent_num = batch.max_clip_num
            else:
                other = cva.synthetic.make_batch(9234 + rank, w["B"], w["C"], w["Lv"], w["Lc"], w["Lp"], w["Ls"], w["Dv"], w["Dt"], ragged=False)
            other.global_max_synced = True
            pair, turn = (batch, other), [0]
        la_on = [lookahead]  # (the roofline leg of hbm_stress also times the input LayerNorm launched the plain way)

        def step():
            if mode in ("native", "native-graph"):  # N > 1: native phases with the RCCL collectives between them
                # back-to-back steps: the text side's update tail overlaps the next step's forward (COOT_STEP_DEFER_TEXT_JOIN); every
                # step is complete when the timed region ends (barrier + device synchronisation below)
                # the data loader's lookahead: the next batch is announced to the step, which runs that batch's parameter-free input
                # LayerNorm next to its global networks — every step still executes one per side
                if two_batches:
                    cur, nxt = pair[turn[0] & 1], pair[(turn[0] + 1) & 1]
                    turn[0] += 1
                    return trainer.train_step_native(cur, vid_counts=vid_counts, clip_counts=clip_counts, defer_join=not args.no_defer_join,
                                                     next_batch=nxt if la_on[0] else None)[0]
                return trainer.train_step_native(batch, vid_counts=vid_counts, clip_counts=clip_counts, defer_join=not args.no_defer_join,
                                                 use_graph=(mode == "native-graph"))[0]
            return trainer.train_step(batch, vid_counts, clip_counts)[0]

    def barrier():
        if dp is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def sysfs_clocks(tag):
        if not (args.sysfs_clocks and rank == 0):
            return
        import glob as _g
        out = []
        for f in sorted(_g.glob("/sys/class/drm/card*/device/pp_dpm_*clk")):
            try:
                cur = [ln.strip() for ln in open(f).read().splitlines() if "*" in ln]
                out.append(f.split("/")[4] + ":" + os.path.basename(f)[7:] + "=" + ",".join(cur))
            except OSError:
                pass
        sys.stderr.write(f"sysfs clocks [{tag}]: " + " ".join(out) + "\n")

    if args.pre_burn:
        kind, ms_b = args.pre_burn.split(":")
        ms_b = float(ms_b)
        # one untimed step first: its lazy initialisation leaves the device idle for ~50 ms, which would undo the burn
        step()
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        if kind == "none":
            pass
        elif kind == "mfma":
            # MFMA-dense GEMMs: tests whether the slow weight-gradient launches of the first steps (tools/rocpd_early_late.py) are the
            # device's power management ramping up under matrix load after an idle period
            a_ = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16); b_ = torch.randn(8192, 8192, device="cuda").to(torch.bfloat16)
            torch.mm(a_, b_); torch.cuda.synchronize()
            tb0 = time.perf_counter()
            while time.perf_counter() - tb0 < ms_b * 1e-3:
                for _ in range(4):
                    torch.mm(a_, b_)
                torch.cuda.synchronize()
            del a_, b_
        elif kind == "hbm":
            a_ = torch.empty(256 << 20, dtype=torch.float32, device="cuda"); b_ = torch.empty_like(a_)
            while time.perf_counter() - tb0 < ms_b * 1e-3:
                for _ in range(8):
                    b_.copy_(a_)
                torch.cuda.synchronize()
            del a_, b_
        else:
            a_ = torch.ones(1 << 20, dtype=torch.float32, device="cuda")
            while time.perf_counter() - tb0 < ms_b * 1e-3:
                for _ in range(200):
                    a_.mul_(1.0001)
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    sysfs_clocks("before warm-up")
    mon_early = None
    if args.clock_monitor_early and rank == 0:
        nsamp_e = int((args.warmup + args.steps) * 1.4 * 10) + 300   # 100 us samples: ~1.4 ms per step + margin
        mon_early = torch.zeros(2 * nsamp_e, dtype=torch.int64, device="cuda")
        side_e = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(side_e):
            cva.lib.check(lib.coot_debug_clock_monitor(mon_early.data_ptr(), nsamp_e, 10000, side_e.cuda_stream), "clock_monitor")
        time.sleep(0.002)  # the first samples are the idle device
    evs, host_t = [], []

    def stamp():
        if args.per_step_events:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
            host_t.append(time.perf_counter())

    stamp()
    for _ in range(args.warmup):
        step()
        stamp()
    barrier()
    if args.idle_ms > 0:
        time.sleep(args.idle_ms * 1e-3)
    sysfs_clocks("after warm-up")
    stamp()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
        stamp()
    host_issue = time.perf_counter() - t0  # CPU time to enqueue K steps (informational)
    barrier()
    elapsed = time.perf_counter() - t0
    sysfs_clocks("after the timed steps")
    blocks = [1e3 * elapsed / args.steps]
    for _ in range(args.repeat - 1):  # regime probe: the same bracket again, in the same process
        tb = time.perf_counter()
        for _ in range(args.steps):
            step()
            stamp()
        barrier()
        blocks.append(1e3 * (time.perf_counter() - tb) / args.steps)
    if rank == 0 and args.repeat > 1:
        sys.stderr.write("blocks of %d steps, ms/step: %s\n" % (args.steps, " ".join(f"{b:.4f}" for b in blocks)))
    if mon_early is not None:
        torch.cuda.synchronize()
        m = mon_early.cpu().numpy().reshape(-1, 2).astype(np.float64)
        ghz = np.diff(m[:, 1]) / (np.diff(m[:, 0]) * 10.0)
        sys.stderr.write("shader clock from before the warm-up, GHz per 1 ms (10 samples of 100 us): " +
                         " ".join(f"{ghz[i:i + 10].mean():.2f}" for i in range(0, len(ghz) - 9, 10)) + "\n")
    if rank == 0 and evs:
        dts = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
        sys.stderr.write("per-step device ms (event to event; warm-up first, then the barrier gap, then the timed steps):\n  " +
                         " ".join(f"{t:.3f}" for t in dts) + "\n")
        sys.stderr.write("per-step HOST ms (enqueue time between the same points):\n  " +
                         " ".join(f"{1e3 * (host_t[i + 1] - host_t[i]):.3f}" for i in range(len(host_t) - 1)) + "\n")
    if dp is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        cdist._all_reduce(t, torch.distributed.ReduceOp.MAX)  # (host-staged under gloo)
        elapsed = float(t.item())
    loss_val = float(last)
    loss_words = None
    if not args.eval and getattr(trainer, "_native", None) is not None and getattr(trainer._native, "losses", None) is not None:
        loss_words = [float(v) for v in trainer._native.losses[:3].tolist()]  # (total, contrastive, cycle-consistency) of the last step
    if args.step_stamps and rank == 0 and not args.eval:
        # (regime probe) the FIRST step after an idle device: the host is not ahead, every launch arrives just in time
        lib.coot_set_option(b"step_stamps", 1)
        torch.cuda.synchronize()
        step()
        buf = C.create_string_buffer(8192)
        lib.coot_debug_step_stamps(buf, 8192)
        lib.coot_set_option(b"step_stamps", 0)
        sys.stderr.write("step timeline of the FIRST step after a device synchronisation (host not ahead):\n" + buf.value.decode())
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        # HIP-event timeline of one more step, enqueued behind two others so the host is ahead as in the timed loop
        lib.coot_set_option(b"step_stamps", 1)
        for _ in range(3):
            step()
        buf = C.create_string_buffer(8192)
        lib.coot_debug_step_stamps(buf, 8192)
        lib.coot_set_option(b"step_stamps", 0)
        sys.stderr.write("step timeline (HIP events, us since the step's first launch):\n" + buf.value.decode())
    ms_per_step = 1e3 * elapsed / args.steps
    value = clip_pairs * args.steps / elapsed
    if args.clock_monitor and rank == 0:
        # shader clock the device delivers under this load (outside the timed region): one wave on a side stream samples the
        # 100 MHz real-time counter and the shader clock counter every 10 us while 8 more steps run
        nsamp = int(8 * ms_per_step * 100) + 200
        mon = torch.zeros(2 * nsamp, dtype=torch.int64, device="cuda")
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            cva.lib.check(lib.coot_debug_clock_monitor(mon.data_ptr(), nsamp, 1000, side.cuda_stream), "clock_monitor")
        time.sleep(0.0005)
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        m = mon.cpu().numpy().reshape(-1, 2).astype(np.float64)
        ghz = np.diff(m[:, 1]) / (np.diff(m[:, 0]) * 10.0)
        busy = ghz[20:int(8 * ms_per_step * 100) - 20]
        sys.stderr.write(f"shader clock under load (10 us samples over 8 steps): mean {busy.mean():.2f} GHz, p10 {np.percentile(busy, 10):.2f}, "
                         f"p90 {np.percentile(busy, 90):.2f}; idle tail {ghz[-50:].mean():.2f} GHz\n")
        per = max(1, int(ms_per_step * 100 / 16))
        sys.stderr.write("  per 1/16 step (GHz, first monitored step): " + " ".join(f"{ghz[20 + i * per:20 + (i + 1) * per].mean():.2f}" for i in range(16)) + "\n")

    roofline = None
    if rank == 0 and not args.no_roofline:
        # same steps again with HIP events around every gemm_nt launch (the dominant kernel: all Linear layers,
        # forward and dX).  Kept out of the timed region so `value` carries no instrumentation overhead.
        nst = max(2, min(5, args.steps))
        lib.coot_timing_enable(1)
        for _ in range(nst):
            step()
        torch.cuda.synchronize()
        names = {2: "gemm_nt_kernel (LDS-staged bf16 MFMA GEMM: Linear fwd + dX of the local networks)",
                 3: "gemm_nt_small_kernel (direct-from-L2 fragments, M <= 512: global networks)",
                 4: "gemm_tn_kernel + gemm_tn_reduce_kernel (weight gradients, split over tokens)",
                 5: "fused token-tile chain kernels (infc_qkv_fwd, post_attn_fwd, pre_attn_bwd, qkv_bwd: LDS-resident 128 x 384 tile, "
                    "384-wide GEMM passes with L2-streamed weights)"}

        def collect(sel):
            ms, fl, n = C.c_double(), C.c_double(), C.c_int()
            cva.lib.check(lib.coot_timing_collect(sel, C.byref(ms), C.byref(fl), C.byref(n)), "timing_collect")
            tf = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
            return {"achieved": round(tf, 2), "frac": round(tf / 2500.0, 4), "launches_per_step": n.value // nst,
                    "avg_launch_us": round(1e3 * ms.value / max(n.value, 1), 2), "ms_per_step": round(ms.value / nst, 3),
                    "algorithmic_gflop_per_step": round(fl.value / nst / 1e9, 2)}

        names[6] = "single-launch global network passes (glob_fwd: LDS-resident 32-row tiles, twelve 384-wide GEMM passes, L2-streamed weights)"
        by = {names[k]: collect(k) for k in (2, 3, 4, 5, 6)}
        # algorithmic HBM bytes of the fused chains: bf16 tensors each read / written exactly once per token:
        # input FC + QKV (reads xhat [Din], writes h0, z0, qkv), forward chain 1536 read + 8448 written, backward chain 5376 + 5376,
        # QKV dX 3840 + 768
        if ragged and not args.padded:  # rows the kernels process: the valid tokens
            tok_v = int(batch.vid_feat_len.sum() + batch.clip_feat_len.sum()); tok_t = int(batch.par_feat_len.sum() + batch.sent_feat_len.sum())
        else:
            tok_v = batch.vid_feat.shape[0] * batch.vid_feat.shape[1] + batch.clip_feat.shape[0] * batch.clip_feat.shape[1]
            tok_t = batch.par_feat.shape[0] * batch.par_feat.shape[1] + batch.sent_feat.shape[0] * batch.sent_feat.shape[1]
        fused = names[5]
        if by[fused]["launches_per_step"]:
            alg = tok_v * (2 * w["Dv"] + 1536 + 2304) + tok_t * (2 * w["Dt"] + 1536 + 2304) + (tok_v + tok_t) * (9984 + 10752 + 4608)
            by[fused]["algorithmic_bytes_per_launch"] = int(alg / by[fused]["launches_per_step"])
        allk, infc = collect(0), collect(1)
        inln = collect(7)  # input LayerNorm: "achieved" is TB/s here (the slot carries algorithmic bytes)
        lib.coot_timing_enable(0)
        inln_plain = None
        if args.workload == "hbm_stress" and not args.eval and la_on[0]:
            # the same kernel launched the plain way (at the head of its own step, one workgroup per four rows, normal loads) next to the
            # prefetched launch above (capped grid, streaming loads, beside the global networks): two different operating points of the
            # HBM-roofline kernel of BASELINE.json configs[4], reported separately
            la_on[0] = False
            step()  # (this batch was still announced by the step before it: no LayerNorm launch here)
            torch.cuda.synchronize()
            lib.coot_timing_enable(1)
            for _ in range(nst):
                step()
            torch.cuda.synchronize()
            inln_plain = collect(7)
            lib.coot_timing_enable(0)
            la_on[0] = True
        dom = max(by, key=lambda k: by[k]["ms_per_step"])  # the kernel the step spends most MFMA time in
        # HBM bytes per launch of that family from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE and
        # WRITE_SIZE collected in separate runs, corrected as MI355X_MICROARCH.md prescribes: tools/pmc_traffic.py)
        traffic, traffic_src, traffic_stale = None, None, None
        fam_key = {2: "gemm_nt", 3: "gemm_nt_small", 4: "gemm_tn", 5: "fused", 6: "glob"}[[k for k in names if names[k] == dom][0]]
        import glob
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, os.path.join(here, "tools"))
        from sources_hash import sources_sha16
        live_sha = sources_sha16(here)
        cands = sorted(glob.glob(os.path.join(here, "profiles", "r*_traffic.json")))
        if cands and args.workload == "anet" and not args.eval:  # (the committed PMC passes are of THIS command: the default workload's train step)
            try:
                tj = json.load(open(cands[-1]))
                traffic_src = os.path.relpath(cands[-1], here)
                if tj.get("kernel_sources_sha16") == live_sha:
                    traffic = tj["families"][fam_key]["hbm_bytes_per_launch"]
                else:
                    # the counters were collected on other kernel sources than this run's: not quoted (a traffic regression must not hide
                    # behind an old profile); re-run tools/profile_round.sh and commit its traffic.json
                    traffic_stale = {"profile_kernel_sources_sha16": tj.get("kernel_sources_sha16"), "this_run_kernel_sources_sha16": live_sha,
                                     "stale_value": tj["families"][fam_key]["hbm_bytes_per_launch"]}
                    sys.stderr.write(f"bench.py: STALE TRAFFIC PROFILE {traffic_src}: taken on kernel sources {tj.get('kernel_sources_sha16')}, this run "
                                     f"is {live_sha}; roofline.traffic = null (tools/profile_round.sh <tag>, then commit profiles/<tag>_traffic.json)\n")
            except (KeyError, ValueError, OSError):
                traffic = None
        # The same family in the committed rocprofv3 kernel trace of this command (profiles/*_kernel_stats_bench_train_anet.csv: sum of the
        # four chain kernels' total time / steps in the trace): a launch's duration there runs from its first workgroup's start to its last
        # one's end under the profiler's own interleaving of the two streams, and differs from the HIP-event figure of the free-running
        # step by how long the text side's 64-row launches wait for CUs (docs/NOTEBOOK_r1-r5.md section 12).  Both are reported; `frac` is the live one.
        trace = None
        if fam_key == "fused" and args.workload == "anet" and not args.eval:
            csvs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_kernel_stats_bench_train_anet.csv")))
            try:
                import csv as _csv
                rows = list(_csv.DictReader(open(csvs[-1])))
                nsteps = next(int(r["calls"]) for r in rows if "sample_idx_kernel" in r["kernel"])
                fam_us = sum(float(r["total_us"]) for r in rows if any(k in r["kernel"] for k in ("infc_qkv_fwd", "post_attn_fwd", "pre_attn_bwd", "qkv_bwd"))) / nsteps
                tf_tr = by[dom]["algorithmic_gflop_per_step"] / fam_us * 1e3  # GFLOP / us = PFLOP/s
                trace = {"frac": round(tf_tr / 2500.0, 4), "achieved": round(tf_tr, 1), "us_per_step": round(fam_us, 1), "steps_in_trace": nsteps,
                         "source": os.path.relpath(csvs[-1], os.path.dirname(os.path.abspath(__file__))),
                         "note": "committed rocprofv3 --kernel-trace of this command (the newest profiles/r*_kernel_stats_bench_train_anet.csv by "
                                 "name), NOT measured in this run"}
            except (IndexError, StopIteration, KeyError, ValueError, OSError, ZeroDivisionError):
                trace = None
        roofline = {"bound": "mfma", "kernel": dom, "achieved": by[dom]["achieved"], "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": by[dom]["frac"], "frac_method": "HIP events around every launch of the family, on its launch stream, in " + str(nst) + " extra steps of this run",
                    "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE, family average; separate "
                    "rocprofv3 --pmc passes of this command on THESE kernel sources: the profile carries their hash, another hash => null)",
                    "traffic_source": traffic_src, **({"traffic_stale": traffic_stale} if traffic_stale else {}),
                    "kernel_sources_sha16": live_sha,
                    "algorithmic_bytes_per_launch": by[dom].get("algorithmic_bytes_per_launch"),
                    # NOT measured in this run: figures read from files committed under profiles/
                    "committed_profile": {"kernel_trace": trace},
                    "launches_per_step": by[dom]["launches_per_step"],
                    "avg_launch_us": by[dom]["avg_launch_us"], "ms_per_step": by[dom]["ms_per_step"],
                    # the same launches against the OTHER roofline: their algorithmic bytes over the same durations (the chains save
                    # 8.4 KB per token for the backward pass; with every CU busy their store phases run at the chip's write bandwidth,
                    # profiles/README.md "Round 3") — two streams share the chip, so neither fraction can approach 1 on its own
                    "hbm_view": ({"achieved": round(by[dom]["algorithmic_bytes_per_launch"] / by[dom]["avg_launch_us"] / 1e3, 1), "peak": 8000.0,
                                  "unit": "GB/s", "frac": round(by[dom]["algorithmic_bytes_per_launch"] / by[dom]["avg_launch_us"] / 1e3 / 8000.0, 4)}
                                 if by[dom].get("algorithmic_bytes_per_launch") and by[dom]["avg_launch_us"] else None),
                    "note": "algorithmic 2*M*N*K per launch / HIP-event duration on the launch stream, measured while both "
                            "sides (two streams) run concurrently; 'by_kernel' lists every MFMA kernel family the same way.  Since round 6 the "
                            "family's post_attn_fwd launches also compute the layer's forward self-attention (formerly two attn_short_fwd launches "
                            "per step, ~135 us of kernel time, HBM bound, 3.6 GFLOP): their durations and FLOPs are in this figure",
                    "all_mfma_kernels": allk, "input_fc_instances": infc, "by_kernel": by,
                    "input_layernorm_hbm": {"achieved_GBps": round(inln["achieved"] * 1e3, 1), "frac_of_8TBps": round(inln["achieved"] / 8.0, 4),
                                            "launches_per_step": inln["launches_per_step"], "avg_launch_us": inln["avg_launch_us"],
                                            "algorithmic_bytes_per_step": int(inln["algorithmic_gflop_per_step"] * 1e9),
                                            "note": "ln_fwd_kernel: reads the fp32 feature stream once, writes bf16 xhat once"}}
        if args.workload == "hbm_stress":  # BASELINE.json configs[4]: the input stream against the HBM roofline
            roofline = {"bound": "hbm", "kernel": "ln_fwd_kernel (input LayerNorm: the kernel that reads the feature stream)",
                        "achieved": round(inln["achieved"] * 1e3, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(inln["achieved"] / 8.0, 4),
                        "traffic": None, "launches_per_step": inln["launches_per_step"], "avg_launch_us": inln["avg_launch_us"],
                        "algorithmic_bytes_per_launch": int(inln["algorithmic_gflop_per_step"] * 1e9 / max(inln["launches_per_step"], 1)),
                        "note": "algorithmic bytes (fp32 features read once + bf16 normalised features written once) / HIP-event duration; "
                                "mfma families under by_kernel", "by_kernel": by, "all_mfma_kernels": allk,
                        "launch": ("prefetched for the NEXT batch inside the step (ln_fwd_stream_kernel: 512 workgroups, streaming loads / stores, beside "
                                   "the global networks)" if la_on[0] else "plain (ln_fwd_kernel at the head of the step)")}
            if inln_plain is not None and inln_plain["launches_per_step"]:
                roofline["plain_launch"] = {"kernel": "ln_fwd_kernel at the head of its own step (--no-lookahead operating point)",
                                            "achieved": round(inln_plain["achieved"] * 1e3, 1), "unit": "GB/s", "frac": round(inln_plain["achieved"] / 8.0, 4),
                                            "avg_launch_us": inln_plain["avg_launch_us"], "launches_per_step": inln_plain["launches_per_step"]}
    if dp is not None:
        torch.distributed.barrier()

    if rank == 0:
        if ragged:
            lens = {"vid": batch.vid_feat_len.cpu().numpy(), "clip": batch.clip_feat_len.cpu().numpy(), "par": batch.par_feat_len.cpu().numpy(),
                    "sent": batch.sent_feat_len.cpu().numpy(), "counts": batch.clip_num.cpu().numpy(), "Dv": w["Dv"], "Dt": w["Dt"]}
            fwd_flops, train_flops = algorithmic_flops_ragged(lens)   # rank 0's shard; x world below
        else:
            fwd_flops, train_flops = algorithmic_flops_per_step(w, cfg)
        out = {
            "metric": "clip-pairs/sec (COOT retrieval " + ("eval forward" if args.eval else "train step") + ", whole job)",
            "value": round(value, 1), "unit": "clip-pairs/s", "n_gpus": (torch.distributed.get_world_size() if dp is not None else 1), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": cva.lib.operand(), "data": "synthetic",  # the loaded build's MFMA operand format (COOT_OPERAND=f16: forward-only, --eval)
            "config": {"workload": f"{cva.synthetic.WORKLOAD_LABEL[args.workload]}: {w['B']} videos x "
                                   f"{(str(w['C']) + ' clips') if not ragged else (str(clip_counts[0]) + ' clips in total (rank 0)')} per GPU, "
                                   f"Lc=Lv={w['Lc']}, Ls={w['Ls']}, Lp={w['Lp']}, Dv={w['Dv']}, Dt={w['Dt']}, d_model=384",
                       "global_batch_videos": w["B"] * world, "clip_pairs_per_step": clip_pairs,
                       "parallelism": f"dp{world}", "mode": "eval" if args.eval else "train",
                       **({"collectives": {"direct": "RCCL calls on the step's own streams (dist.DirectRccl)", "torch": "torch.distributed"}[dp.collectives_route()]}
                          if (dp is not None and not args.eval) else {}),
                       "launch": "eval" if args.eval else mode,
                       **({"batches": "two different synthetic batches in turn"} if (not args.eval and two_batches) else {}),
                       **({"input_lookahead": "each step runs the NEXT batch's input LayerNorm (one per side per step, as without it) next to its global networks"}
                          if (not args.eval and lookahead and getattr(getattr(trainer, "_native", None), "stages", None) is not None) else {}),
                       **({"token_layout": "padded to the batch maxima (the reference's layout)" if args.padded else "packed (cu_seqlens): valid tokens only",
                           "valid_tokens": [int(batch.vid_feat_len.sum() + batch.clip_feat_len.sum()), int(batch.par_feat_len.sum() + batch.sent_feat_len.sum())],
                           "padded_tokens": [batch.vid_feat.shape[0] * batch.vid_feat.shape[1] + batch.clip_feat.shape[0] * batch.clip_feat.shape[1],
                                             batch.par_feat.shape[0] * batch.par_feat.shape[1] + batch.sent_feat.shape[0] * batch.sent_feat.shape[1]]}
                          if ragged else {}),
                       "final_loss": round(loss_val, 5), **({"final_losses": loss_words} if loss_words else {}),
                       **({"share_device": True, "note": "TEST MODE: all ranks share device 0 through gloo (host-staged collectives); not a scaling number"}
                          if args.share_device else {})},
            "per_gpu": round(value / world, 1), **({"blocks_ms_per_step": [round(b, 4) for b in blocks]} if args.repeat > 1 else {}),
            "host_issue_ms_per_step": round(1e3 * host_issue / args.steps, 3),
            "algorithmic_tflops_per_s": round((fwd_flops if args.eval else train_flops) * world / (ms_per_step * 1e-3) / 1e12, 2),
        }
        if roofline is not None:
            out["roofline"] = roofline
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if dp is not None:
        dp.close()  # (the direct RCCL communicator of the step's collectives)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
