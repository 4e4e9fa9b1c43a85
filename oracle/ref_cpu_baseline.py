"""Time the REAL reference (``/root/reference/openrl``) on CPU at BASELINE.json configs[1] - TEST INFRASTRUCTURE.

``cpu_baseline.kind = "reference"`` of SURVEY.md section 8d: the reference's own ``PPOModule`` +
``NormalReplayBuffer`` + ``PPOAlgorithm`` (``algorithms/ppo.py``, ``buffers/replay_data.py``,
``modules/ppo_module.py``) under the import stubs of ``oracle/ref_stubs.py``, driven by the restated driver loop
(``drivers/onpolicy_driver.py:57-279`` / ``rl_driver.py:118-169`` - the drivers themselves need gymnasium, which is
not installable offline) on the same synthetic fixed-step env as ``bench.py``, same shapes and defaults
(4096 envs x 128 steps, obs 4, Discrete(2), ppo_epoch 10, one minibatch, ValueNorm on), 1 warm-up + 3 iterations,
wall clock with ``time.perf_counter``.

Runs only where ``/root/reference`` exists (the authoring container; the GPU box has no reference):

    python -m oracle.ref_cpu_baseline [--threads N] [--iters 3] [--warmup 1] [--out profiles/r02_ref_cpu_line.json]

``bench.py`` cites the committed line next to its own ``"kind": "port"`` measurement of ``oracle/cpu_trainer.py``
(the restatement of the same loop, which DOES travel to the GPU box).
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch

from . import ppo_oracle as po
from . import ref_stubs


def build(n_envs: int, T: int, obs_dim: int, n_act: int, ppo_epoch: int, seed: int = 0):
    ref_stubs.install()
    from gymnasium.spaces import Box, Discrete
    from openrl.algorithms.ppo import PPOAlgorithm
    from openrl.buffers import NormalReplayBuffer
    from openrl.modules.ppo_module import PPOModule
    from openrl.utils.util import set_seed

    cfg = ref_stubs.reference_cfg(["--ppo_epoch", str(ppo_epoch), "--num_mini_batch", "1", "--episode_length", str(T)])
    cfg.num_agents, cfg.n_rollout_threads, cfg.learner_n_rollout_threads = 1, n_envs, n_envs
    cfg.rnn_hidden_size, cfg.episode_length, cfg.seed = cfg.hidden_size, T, seed
    obs_space, act_space = Box(-np.inf, np.inf, (obs_dim,)), Discrete(n_act)
    set_seed(cfg.seed)
    module = PPOModule(cfg, policy_input_space=obs_space, critic_input_space=obs_space, act_space=act_space,
                       share_model=False, rank=0, world_size=1)
    buffer = NormalReplayBuffer(cfg, 1, obs_space, act_space, data_client=None)
    algo = PPOAlgorithm(cfg, module, agent_num=1)
    env = po.SynthEnvOracle(n_envs, obs_dim, seed, 200)
    return cfg, module, buffer, algo, env


def iterate(cfg, module, buffer, algo, env, phase):
    """One ``OnPolicyDriver._inner_loop`` (onpolicy_driver.py:57-78): actor_rollout (:154-203) with act (:235-279) and
    add2buffer (:80-152), compute_returns (:205-233), ``PPOAlgorithm.train``, ``buffer.after_update``."""
    N, T = cfg.n_rollout_threads, cfg.episode_length
    d = buffer.data
    algo.prep_rollout()
    for step in range(T):
        t0 = time.perf_counter()
        with torch.no_grad():
            value, action, logp, rs_a, rs_c = module.get_actions(
                d.get_batch_data("critic_obs", step), d.get_batch_data("policy_obs", step),
                d.get_batch_data("rnn_states", step), d.get_batch_data("rnn_states_critic", step),
                d.get_batch_data("masks", step), action_masks=d.get_batch_data("action_masks", step))
        split = lambda x: np.array(np.split(x.detach().cpu().numpy(), N))
        values, actions, logps, rnn_a, rnn_c = split(value), split(action), split(logp), split(rs_a), split(rs_c)
        t1 = time.perf_counter()
        obs, rewards, dones, infos = env.step(actions)
        t2 = time.perf_counter()
        dones_env = np.all(dones, axis=1)
        rnn_a[dones_env] = np.zeros((dones_env.sum(), 1, cfg.recurrent_N, cfg.hidden_size), dtype=np.float32)
        rnn_c[dones_env] = np.zeros((dones_env.sum(), 1, cfg.recurrent_N, cfg.hidden_size), dtype=np.float32)
        masks = np.ones((N, 1, 1), dtype=np.float32)
        masks[dones_env] = np.zeros((dones_env.sum(), 1, 1), dtype=np.float32)
        active = np.ones((N, 1, 1), dtype=np.float32)
        active[dones] = np.zeros((dones.sum(), 1), dtype=np.float32)
        active[dones_env] = np.ones((dones_env.sum(), 1, 1), dtype=np.float32)
        bad = np.array([[[0.0] if "bad_transition" in info and info["bad_transition"][a] else [1.0] for a in range(1)]
                        for info in infos])
        buffer.insert(obs, rnn_a, rnn_c, actions, logps, values, rewards, masks, active_masks=active, bad_masks=bad,
                      action_masks=None)
        t3 = time.perf_counter()
        phase["act"] += t1 - t0
        phase["env"] += t2 - t1
        phase["insert"] += t3 - t2
    t0 = time.perf_counter()
    with torch.no_grad():
        nv = module.get_values(d.get_batch_data("critic_obs", -1), np.concatenate(d.rnn_states_critic[-1]),
                               np.concatenate(d.masks[-1]))
    next_values = np.array(np.split(nv.detach().cpu().numpy(), N))
    buffer.compute_returns(next_values, module.get_critic_value_normalizer())
    t1 = time.perf_counter()
    algo.prep_training()
    info = algo.train(d)
    buffer.after_update()
    t2 = time.perf_counter()
    phase["gae"] += t1 - t0
    phase["update"] += t2 - t1
    return info


def run(n_envs=4096, T=128, obs_dim=4, n_act=2, ppo_epoch=10, iters=3, warmup=1, threads=None):
    threads = threads or os.cpu_count()
    torch.set_num_threads(threads)
    cfg, module, buffer, algo, env = build(n_envs, T, obs_dim, n_act, ppo_epoch)
    buffer.init_buffer(env.reset())
    phase = {k: 0.0 for k in ("act", "env", "insert", "gae", "update")}
    for _ in range(warmup):
        iterate(cfg, module, buffer, algo, env, phase)
    phase = {k: 0.0 for k in phase}
    t0 = time.perf_counter()
    info = {}
    for _ in range(iters):
        info = iterate(cfg, module, buffer, algo, env, phase)
    dt = time.perf_counter() - t0
    return {"kind": "reference", "value": n_envs * T * iters / dt, "unit": "env-steps/s", "cores": threads,
            "host_cpus": os.cpu_count(), "seconds": dt, "iters": iters, "warmup": warmup,
            "workload": "configs[1]: %d envs x %d steps, obs %d, Discrete(%d), ppo_epoch %d, 1 minibatch, ValueNorm on; "
                        "synthetic fixed-step env" % (n_envs, T, obs_dim, n_act, ppo_epoch),
            "objects": "openrl.modules.ppo_module.PPOModule + openrl.buffers.NormalReplayBuffer + "
                       "openrl.algorithms.ppo.PPOAlgorithm (reference v0.2.1, torch %s CPU)" % torch.__version__,
            "phase_seconds": {k: round(v, 3) for k, v in phase.items()},
            "last_train_info": {k: float(v) for k, v in info.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    res = run(n_envs=args.envs, T=args.steps, iters=args.iters, warmup=args.warmup, threads=args.threads)
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(line + "\n")


if __name__ == "__main__":
    main()
