#!/usr/bin/env python
"""bench.py - env-steps/sec (collect + GAE + PPO update) of the MI355X engine on BASELINE.json's
config 2: CartPole-shaped PPO, 4096 envs x 128-step rollout, obs 4, Discrete(2), hidden 64,
ppo_epoch 10, num_mini_batch 1, ValueNorm on (reference defaults).

A "step" = ONE training iteration of the hot path over one batch: a 128-step rollout of the 4096 envs on the
synthetic fixed-step env + bootstrap value + GAE/advantages + 10 PPO epochs
(forward, loss, backward, grad-clip, Adam for both towers).

Multi-GPU = the metric's "4096-env CartPole at 1/2/4/8 GPUs" (SURVEY.md section 8e): STRONG scaling - the 4096 global
envs are sharded 4096/G per rank (one process per GPU; rollout, buffer shard and GAE are local), gradients /
statistics are summed over ranks once per optimiser step by the one-shot xGMI all-reduce fused into the
optimiser-step launches (``orl_comm``; ``--collective rccl`` = one RCCL all-reduce instead).  ``--scaling weak`` keeps
4096 envs on EVERY rank.

    python bench.py [--gpus N --steps K --warmup W]
        (N>1: spawns its own N ranks through torch.distributed.run, or joins the group torchrun already made)

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel pair (``orl_ppo_fwd_bwd`` =
ppo_tower_kernel policy + critic, fp32 MFMA bound) timed live with HIP events on the launch stream;
``cpu_baseline`` times the oracle port of the reference's CPU path (oracle/cpu_trainer.py) on the host
cores and cites the committed run of the REAL reference objects (oracle/ref_cpu_baseline.py,
profiles/r04_ref_cpu_line.json) - a reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_ENVS, T_ROLL, OBS_DIM, N_ACT, PPO_EPOCH = 4096, 128, 4, 2, 10
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak (dense)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak (same guide)
HBM_PEAK_GBS = 8000.0


def self_launch(n: int) -> int:
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and the hipIpc inboxes of orl_comm need it
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def link_type_rank0_rank1():
    """Link type between the devices of ranks 0 and 1 as `rocm-smi --showtopotype` prints it (XGMI / PCIE), or a reason."""
    import subprocess

    try:
        if torch.cuda.device_count() < 2:
            return "same device (ranks share the one visible GPU)"
        txt = subprocess.run(["rocm-smi", "--showtopotype"], capture_output=True, text=True, timeout=30).stdout
        rows = [ln.split() for ln in txt.splitlines() if ln.strip().startswith("GPU")]
        hdr = next((r for r in rows if len(r) > 1 and all(c.startswith("GPU") for c in r)), None)
        row0 = next((r for r in rows if r[0] == "GPU0" and not all(c.startswith("GPU") for c in r)), None)
        if hdr and row0 and "GPU1" in hdr:
            return row0[1 + hdr.index("GPU1")]
        return "unparsed: " + " | ".join(txt.splitlines()[:12])[:400]
    except Exception as e:  # rocm-smi missing / hung: report, never fail the bench over it
        return "unavailable: %s" % (e,)


def multi_gpu_report(trainer, args, dev, world, rank, make_engine):
    """Outside the timed region, world > 1 only (round-3 VERDICT item 4a): what the first contact with real xGMI links should
    tell - latency of the one-shot all-reduce and of one RCCL all-reduce at the optimiser step's message size, which
    collective the timed run used and how the comm's self-test went, the weak-scaling rate next to the strong one, the link
    type.  Raises (rc != 0) if any rank's comm error word is set."""
    from openrl_amd import distributed as du

    rep = {"collective_setup": dict(du.LAST_SETUP), "link_rank0_rank1": link_type_rank0_rank1(),
           "backend": torch.distributed.get_backend()}
    n = int(trainer._sums.numel())
    rep["message_bytes"] = 4 * n
    reps = 1000

    def timed(fn):
        torch.distributed.barrier()
        torch.cuda.synchronize()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        torch.distributed.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        t = torch.tensor([us], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return round(float(t.item()), 2)

    # Every rank runs the SAME sequence of collectives below whatever happens locally (a diagnostic that throws on one rank
    # only would leave the others inside a collective): failures are recorded as strings, never raised - except the comm's
    # error word, which must fail the run.
    vec = torch.zeros(n, dtype=torch.float32, device=dev)
    comm = getattr(trainer, "_comm", None)
    if comm is not None:
        try:
            rep["orl_allreduce_small_us"] = timed(lambda: comm.allreduce_(vec))
        except Exception as e:  # noqa: BLE001
            rep["orl_allreduce_small_us"] = "error: %s" % (e,)
    else:
        rep["orl_allreduce_small_us"] = None
    try:
        rep["torch_all_reduce_us"] = timed(lambda: torch.distributed.all_reduce(vec))
    except Exception as e:  # noqa: BLE001
        rep["torch_all_reduce_us"] = "error: %s" % (e,)
    rep["latency_note"] = ("%d back-to-back calls, host clock around the loop, max over ranks; the fused path carries the same "
                           "exchange INSIDE the reduce/apply launches (no extra launch)" % reps)
    # the comm's device error word on every rank: one MAX all-reduce, then fail loudly
    err = comm.error_flag().clone().to(torch.int32) if comm is not None else torch.zeros(1, dtype=torch.int32, device=dev)
    torch.distributed.all_reduce(err, op=torch.distributed.ReduceOp.MAX)
    rep["comm_error_word_max_over_ranks"] = int(err.item())
    if int(err.item()) != 0:
        raise SystemExit("orl_comm error word set on some rank (%d): a peer's contribution timed out - the timed run summed "
                         "partial gradients" % int(err.item()))
    if args.scaling == "strong" and not args.no_weak_leg:
        # (never raises between collectives: every local stage's outcome is agreed on by all ranks, see _weak_leg)
        rep["weak_scaling"] = _weak_leg(args, dev, world, make_engine, first_comm=comm)
    return rep


def _all_ok(local_ok, dev):
    """One MIN all-reduce of a status flag: every rank learns whether EVERY rank got through its last local stage."""
    t = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    return int(t.item()) == 1


def _weak_leg(args, dev, world, make_engine, first_comm=None):
    """Weak scaling next to the strong number: the full 4096 envs on EVERY rank, a short run of the same loop.

    Local failures never skip a collective: after each local stage (engine construction, warm-up, timed loop) the ranks
    agree on its outcome with one status all-reduce (`_all_ok`) and either all go on or all return the error record.  The
    strong leg's comm is closed first, so that only one set of peer mappings is open at a time."""
    if first_comm is not None:
        try:
            first_comm.close()
        except Exception:  # noqa: BLE001
            pass
    why = None
    drv_w = trainer_w = None
    try:
        drv_w, trainer_w = make_engine(args.envs)  # (its comm set-up is itself collective and never raises in between)
    except Exception as e:  # noqa: BLE001
        why = "make_engine: %s: %s" % (type(e).__name__, e)
    if not _all_ok(why is None, dev):
        return {"error": why or "a peer failed in make_engine"}
    k = max(2, min(args.steps, 8))
    try:
        for i in range(2):
            drv_w.episode = i
            drv_w._inner_loop()
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        why = "warm-up: %s: %s" % (type(e).__name__, e)
    if not _all_ok(why is None, dev):
        return {"error": why or "a peer failed in the warm-up"}
    torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        for i in range(k):
            drv_w.episode = 2 + i
            drv_w._inner_loop()
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        why = "timed loop: %s: %s" % (type(e).__name__, e)
    if not _all_ok(why is None, dev):
        return {"error": why or "a peer failed in the timed loop"}
    torch.distributed.barrier()
    torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    err = 0
    if getattr(trainer_w, "_comm", None) is not None:
        try:
            trainer_w._comm.check()
        except Exception:  # noqa: BLE001
            err = 1
    if not _all_ok(err == 0, dev):
        return {"error": "orl_comm error word set on some rank during the weak leg"}
    return {"value": round(args.envs * world * T_ROLL * k / float(t.item()), 1), "unit": "env-steps/s",
            "envs_per_gpu": args.envs, "steps": k, "ms_per_step": round(float(t.item()) / k * 1e3, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--perm", default="device", choices=["device", "reference", "identity"],
                    help="minibatch permutation source; 'reference' = host torch.randperm (bit-exact stream)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: 4096 GLOBAL envs sharded over the ranks (the metric); weak: 4096 envs per rank")
    ap.add_argument("--collective", default="p2p", choices=["p2p", "rccl"],
                    help="gradient exchange: fused one-shot xGMI push all-reduce (orl_comm) or one RCCL all-reduce")
    ap.add_argument("--envs", type=int, default=N_ENVS, help="global env count (default: the metric's 4096)")
    ap.add_argument("--env", default="synthetic", choices=["synthetic", "cartpole"],
                    help="synthetic = the metric's fixed-step env of the named shape (the headline); cartpole = the same "
                         "configuration on the device CartPole-v1 physics (a measurement beside the headline)")
    ap.add_argument("--rollout-kernel", default="chain", choices=["chain", "lockstep"],
                    help="cfg.amd_rollout_kernel: chain = round 6's policy-only step chain + batched critic sweep (default), "
                         "lockstep = the round-5 kernel (comparison)")
    ap.add_argument("--optim-step", default="two_launch", choices=["two_launch", "step", "fused"],
                    help="cfg.amd_optim_step: two_launch = reduce + apply launches (default); step = orl_ppo_step, one launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--tower-gemm", default="split", choices=["split", "fp32"],
                    help="A/B: form the tower's GEMMs on the fp32 MFMA instead of the bf16x3 split (same results, slower)")
    ap.add_argument("--no-weak-leg", action="store_true",
                    help="world > 1: skip the short weak-scaling run (4096 envs per rank) reported next to the strong number")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the few untimed iterations of BASELINE.json's other single-GPU configs (other_configs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: re-launch as N ranks, one process per GPU (torch.distributed.run is what
        # the driver's own N>1 command line uses); rank 0's JSON line passes through on stdout
        return self_launch(args.gpus)

    from openrl_amd import distributed as du
    from openrl_amd import ops
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    local_rank = du.init_from_env()
    world = du.world_size()
    rank = du.rank()
    if world != args.gpus:  # never a silent 1-rank run under an N-GPU label (or the other way round)
        raise SystemExit("--gpus %d but the process group has %d rank(s)" % (args.gpus, world))
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)

    if args.scaling == "strong":  # the metric: the global envs shard contiguously, N/G per GPU
        lo, hi = du.shard_range(args.envs, rank, world)
        n_local = hi - lo
    else:
        n_local = args.envs
    if n_local < 1:
        raise SystemExit("%d envs cannot be sharded over %d ranks" % (args.envs, world))
    class _Agent:
        num_time_steps = 0

    def make_engine(n_envs_local):
        cfg_e = default_cfg(["--episode_length", str(T_ROLL), "--ppo_epoch", str(PPO_EPOCH), "--amd_perm_mode", args.perm,
                             "--log_interval", "1000000", "--amd_collective", args.collective, "--amd_tower_gemm",
                             args.tower_gemm, "--amd_rollout_kernel", args.rollout_kernel, "--amd_optim_step",
                             args.optim_step])
        if args.env == "cartpole":
            env_e = make("CartPole-v1", env_num=n_envs_local, device=dev, seed=cfg_e.seed + 10086 * rank)
        else:
            env_e = make("SyntheticFixedStep-v0", env_num=n_envs_local, obs_dim=OBS_DIM, episode_limit=200, device=dev,
                         seed=cfg_e.seed + 10086 * rank)
        net_e = PPONet(env_e, cfg=cfg_e, device=dev, n_rollout_threads=n_envs_local)
        cfg_e.num_env_steps = n_envs_local * T_ROLL * (args.steps + args.warmup + 16)
        trainer_e = PPOAlgorithm(cfg_e, net_e.module, agent_num=1, device=dev)
        buf_e = NormalReplayBuffer(cfg_e, 1, env_e.observation_space, env_e.action_space, device=dev)
        drv_e = OnPolicyDriver({"cfg": cfg_e, "num_agents": 1, "run_dir": None, "envs": env_e, "device": dev}, trainer_e,
                               buf_e, _Agent(), rank=rank, world_size=world)
        assert drv_e.fused
        drv_e.reset_and_buffer_init()
        drv_e._bench_buf = buf_e
        return drv_e, trainer_e

    drv, trainer = make_engine(n_local)
    buf = drv._bench_buf

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        drv.episode = i
        drv._inner_loop()
    # live timing of the dominant kernel pair with HIP events on the launch stream (torch current stream)
    # HIP events perturb what they measure: every record is a marker packet on the launch stream (~4 us of queue time), and
    # 22 of them per iteration cost 1.3 % of the iteration at 4096 envs and 10 % at a 512-env shard
    # (tools/host_rate_probe.py: the same loop without events).  The kernels are therefore timed on a SAMPLE of the timed
    # steps (every `stride`-th step: all of its tower-pair launches and its GAE launch); roofline.launches_timed says how many.
    tower_events = []
    gae_events = []
    ev_stride = max(1, args.steps // 4)
    orig_cr = buf.data.compute_returns

    def timed_compute_returns(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_cr(*a, **k)
        e1.record()
        gae_events.append((e0, e1))

    # the line's own spread: events at the boundaries of (up to) four blocks of the timed steps - marker packets on the launch
    # stream, read after the run; `value` stays total steps / total time
    n_blocks = min(4, args.steps)
    block_edges = [round(k * args.steps / n_blocks) for k in range(n_blocks + 1)]
    block_events = []
    barrier()
    t0 = time.perf_counter()
    info = {}
    for i in range(args.steps):
        if i in block_edges:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            block_events.append(e)
        sampled = i % ev_stride == 0
        trainer.profile_events = tower_events if sampled else None
        buf.data.compute_returns = timed_compute_returns if sampled else orig_cr
        drv.episode = args.warmup + i
        drv._inner_loop()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    block_events.append(e)
    trainer.profile_events = None
    buf.data.compute_returns = orig_cr
    barrier()
    dt = time.perf_counter() - t0
    block_ms = [block_events[k].elapsed_time(block_events[k + 1]) / max(block_edges[k + 1] - block_edges[k], 1)
                for k in range(len(block_events) - 1)]
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    if getattr(trainer, "_comm", None) is not None:
        trainer._comm.check()  # a peer that never arrived (device-side timeout) must fail the run, not shade the number
    global_envs = args.envs if args.scaling == "strong" else args.envs * world
    steps_total = global_envs * T_ROLL * args.steps
    value = steps_total / dt

    # ---- roofline of the dominant kernel pair
    M = n_local * T_ROLL  # rows of one tower-pair launch on this rank
    f_fwd = 2 * ((OBS_DIM + OBS_DIM) * 64 + 2 * 64 * 64 + 64 * (N_ACT + 1))  # 17 792 flop / row
    flops_per_launch = 3 * f_fwd * M                                          # fwd + dgrad + wgrad, both towers
    ev = tower_events
    k_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    # the same launches in ONE iteration that starts from an idle GPU (after the barrier, outside the timed region): the
    # kernel runs ~3 % faster there than in the middle of a back-to-back run (clocks under sustained load) - reported next to
    # the in-run figure, never as `achieved`
    iso_events = []
    trainer.profile_events = iso_events
    drv.episode = args.warmup + args.steps
    drv._inner_loop()
    trainer.profile_events = None
    barrier()
    iso_ms = sum(a.elapsed_time(b) for a, b in iso_events) / max(len(iso_events), 1)
    achieved_tf = flops_per_launch / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    gae_ms = sum(a.elapsed_time(b) for a, b in gae_events) / max(len(gae_events), 1)
    gae_bytes = 16 * M  # S_gae: 3 reads + 1 write per sample
    # HBM bytes per launch: NOT measured in this process (PMC needs rocprofv3) - read from the committed summary of the
    # separate --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, and only quoted for the shape it was taken at
    traffic, traffic_source, traffic_reason = None, None, None
    for name in ("r06_pmc_hbm.json", "r05_pmc_hbm.json", "r04_pmc_hbm.json", "r03_pmc_hbm.json", "r02_pmc_hbm.json", "r01_pmc_hbm.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                blob = json.load(fh)["orl_ppo_fwd_bwd_pair"]
        except Exception:
            continue
        if M == N_ENVS * T_ROLL:
            traffic = blob.get("hbm_bytes_per_launch", blob.get("hbm_bytes_per_launch_raw"))
            traffic_source = "profiles/" + name
        else:
            traffic_reason = ("the committed PMC passes (profiles/%s) were taken at the full-size launch (%d rows); this "
                              "launch has %d rows" % (name, N_ENVS * T_ROLL, M))
        break
    else:
        traffic_reason = "no committed PMC summary under profiles/"
    # pipe occupancy of the same kernel from the committed counter passes (tools/pmc_tower.sh -> profiles/rNN_pmc_tower.txt):
    # MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), VALU-busy = SQ_ACTIVE_INST_VALU (quad-cycles) x 4
    # / (4 x SQ_BUSY_CU_CYCLES); only quoted for the shape and GEMM path they were taken at
    busy = {}
    for name in ("r06_pmc_tower.txt", "r05_pmc_tower.txt", "r04_pmc_tower.txt", "r03_pmc_tower.txt"):
        try:
            ctr = {}
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                for line in fh:
                    f = line.split()
                    if len(f) >= 2 and f[0].startswith("SQ_"):
                        ctr[f[0]] = float(f[1])
            if M == N_ENVS * T_ROLL and args.tower_gemm == "split" and ctr.get("SQ_BUSY_CU_CYCLES"):
                cu = 4.0 * ctr["SQ_BUSY_CU_CYCLES"]
                busy = {"mfma_busy_frac": round(ctr["SQ_VALU_MFMA_BUSY_CYCLES"] / cu, 4),
                        "valu_busy_frac": round(4.0 * ctr["SQ_ACTIVE_INST_VALU"] / cu, 4),
                        "busy_source": "profiles/" + name}
            break
        except Exception:
            continue
    from openrl_amd import _native as nat

    split_terms = int(nat.load().orl_tower_split_terms())
    n_prod = 3.0 if split_terms == 2 else 6.0
    roofline = {"kernel": "orl_ppo_fwd_bwd (ppo_tower_pair_kernel: policy + critic towers in one launch)", "bound": "mfma",
                "achieved": round(achieved_tf, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved_tf / F32_MFMA_PEAK_TFLOPS, 4),
                "frac_label": "fp32-equivalent: ALGORITHMIC fp32 flops / the fp32 MFMA peak (SURVEY 8d's pricing); with "
                              "tower_gemm=split the products run on the 16-bit MFMA pipe - mfma16_frac / mfma_busy_frac / "
                              "valu_busy_frac are the utilisation figures",
                "traffic": traffic, "traffic_source": traffic_source, "traffic_reason": traffic_reason,
                "launch_ms": round(k_ms, 4), "launches_timed": len(ev),
                "launch_sampling": "HIP events around every tower-pair launch of every %d-th timed step (an event record is a "
                                   "marker packet: 22 per iteration cost 1.3 %% of the iteration)" % ev_stride,
                "launch_ms_from_idle": round(iso_ms, 4),
                "flops_per_launch": flops_per_launch,
                # how the flops are executed: the three 64 x 64 GEMMs of a tile (92 % of the algorithmic flops) run as 6 of
                # the 9 bf16 products of three-term bf16 splits of both operands (the splits are exact, the three smallest
                # cross terms are dropped: ~2^-24 relative), fp32 accumulation - measured error <= the fp32 MFMA's own (profiles/r03_split_bf16_gemm.txt); `frac` stays priced on ALGORITHMIC fp32
                # flops against the fp32 MFMA peak, `mfma16_frac` is the share of the dense 16-bit MFMA peak issued
                # round 6: two-term fp16 splits (hi = rn16(x), lo = rn16(x - hi): 22 significand bits), 3 of the 4 products, every
                # operand scaled by an exact power of two chosen from its own maximum (fp16's range) - the same distance to float64
                # as the three-term bf16 build and the fp32 MFMA (profiles/r06_experiments.md section 8); split_terms says which
                # build is loaded (orl_tower_split_terms)
                "mfma_path": (("fp16x2 split (3 of 4 products, power-of-two operand scaling) on v_mfma_f32_16x16x32_f16 / 32x32x16_f16"
                               if split_terms == 2 else
                               "bf16x3 split (6 of 9 products) on v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16") +
                              "; fc1 + head on v_mfma_f32_16x16x4_f32") if args.tower_gemm == "split" else "v_mfma_f32_16x16x4_f32 (--tower-gemm fp32)",
                "split_terms": split_terms if args.tower_gemm == "split" else None,
                "mfma16_frac": round(n_prod * (16384.0 / 17792.0) * achieved_tf / BF16_MFMA_PEAK_TFLOPS, 4)
                if args.tower_gemm == "split" else 0.0,
                # the pipe the split GEMMs actually run on: n_prod 16-bit products per fp32 product -> its ceiling in
                # fp32-equivalent flops is the dense fp16 / bf16 MFMA peak / n_prod (833 TFLOP/s at 3, 417 at 6)
                "frac_of_split_ceiling": round(achieved_tf / (BF16_MFMA_PEAK_TFLOPS / n_prod), 4)
                if args.tower_gemm == "split" else None,
                # (round 5's key, kept: the ceiling of the split the LOADED build uses - / 3 for the fp16 pairs, / 6 for bf16 x 3)
                "frac_of_bf16_split_ceiling": round(achieved_tf / (BF16_MFMA_PEAK_TFLOPS / n_prod), 4)
                if args.tower_gemm == "split" else None,
                **busy,
                "gae_scan": {"bound": "hbm", "achieved": round(gae_bytes / (gae_ms * 1e-3) / 1e9, 2) if gae_ms else 0,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "launch_ms": round(gae_ms, 4),
                             "note": "%.1f MB per launch: latency-bound, includes adv statistics" % (gae_bytes / 1e6)}}

    out = {"metric": "env-steps/sec (collect+PPO update), 4096-env CartPole-shape", "value": round(value, 1),
           "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 4),
           "ms_per_step_min": round(min(block_ms), 4) if block_ms else None,
           "ms_per_step_max": round(max(block_ms), 4) if block_ms else None,
           "ms_per_step_blocks": "device time per step of %d consecutive blocks of the timed steps (HIP events at the block "
                                 "boundaries): the line's own spread" % len(block_ms),
           "higher_is_better": True, "scaling": args.scaling,
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "dtype_note": "fp32 storage and accumulation everywhere; the 64-wide GEMM products at fp32 ACCURACY on the 16-bit "
                         "MFMA: 3 of the 4 products of two-term fp16 splits (22 significand bits per operand, operands scaled by "
                         "exact powers of two; measured error <= v_mfma_f32_16x16x4_f32's: tools/split_f16_gemm.hip, "
                         "profiles/r06_experiments.md section 8)",
           "config": {"workload": "configs[1]: PPO, %d global envs x 128-step rollout (%d envs per GPU), obs 4, "
                                  "Discrete(2), MLP 64x64, ppo_epoch 10, num_mini_batch 1, ValueNorm on; %s" %
                                  (global_envs, n_local, "synthetic fixed-step env" if args.env == "synthetic" else
                                   "device CartPole-v1 physics (NOT the headline workload)"),
                      "rollout_kernel": args.rollout_kernel, "optim_step": args.optim_step,
                      "global_envs": global_envs, "envs_per_gpu": n_local, "rollout_len": T_ROLL,
                      "ppo_epoch": PPO_EPOCH, "perm_mode": args.perm, "parallelism": "env-shard dp%d" % world,
                      "collective": ("none" if world == 1 else
                                     "orl_comm one-shot xGMI push, fused into reduce+apply" if trainer._comm is not None
                                     else "torch.distributed all_reduce (%s)" % torch.distributed.get_backend())},
           "roofline": roofline}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_trainer import time_cpu_baseline_bounded

        cb = time_cpu_baseline_bounded(n_envs=N_ENVS, ppo_epoch=PPO_EPOCH, target_seconds=args.cpu_seconds)
        ref_run = None
        try:  # the REAL reference objects, timed where /root/reference exists (oracle/ref_cpu_baseline.py), committed
            with open(os.path.join(ROOT, "profiles", "r04_ref_cpu_line.json")) as fh:
                ref = json.loads(fh.readline())
            ref_run = {"value": round(ref["value"], 1), "unit": "env-steps/s", "cores": ref["cores"],
                       "sample": "1 warm-up + %d iterations of the full configs[1] workload with the reference's own "
                                 "classes, timed in the build container" % ref["iters"],
                       "source": "profiles/r04_ref_cpu_line.json"}
        except Exception:
            pass
        if ref_run is not None:  # top-level scalars: the REAL reference classes' rate survives any truncation of the object
            out["cpu_reference_value"] = ref_run["value"]
            out["cpu_reference_cores"] = ref_run["cores"]
        out["cpu_baseline"] = {"value": round(cb["env_steps_per_s"], 1), "unit": "env-steps/s", "cores": cb["cores"],
                               "kind": "port", "reference_run": ref_run,
                               "sample": "1 iteration of the oracle port: %d envs x %d-step rollout + %d full-batch "
                                         "epochs (%.1f s; act %.1f s, insert %.1f s, update %.1f s)"
                                         % (N_ENVS, cb["T"], PPO_EPOCH, cb["seconds"], cb["phase_act"],
                                            cb["phase_insert"], cb["phase_update"])}
    if rank == 0 and world == 1 and not args.no_other_configs:
        # BASELINE.json's other single-GPU configs, a few iterations each, AFTER and outside the timed region
        from benchmarks.other_configs import run_all

        out["other_configs"] = run_all(steps=3, warmup=2, dev=dev)
    if world > 1:
        rep = multi_gpu_report(trainer, args, dev, world, rank, make_engine)  # collective on every rank; rank 0 prints it
        out["multi_gpu"] = rep
        out["scaling_note"] = ("per-rank work is 1/%d of the 4096 global envs; no multi-GPU hardware curve has been measured "
                               "by the builder - this line is the measurement" % world)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
