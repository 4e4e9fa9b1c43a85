import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from openrl_amd.algorithms.ppo import PPOAlgorithm
from openrl_amd.buffers import NormalReplayBuffer
from openrl_amd.configs.config import default_cfg
from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
from openrl_amd.envs.common import make
from openrl_amd.modules.common import PPONet
dev, N, T = "cuda:0", 1024, 25
rec = sys.argv[1] if len(sys.argv) > 1 else "true"
extra = sys.argv[2:]  # e.g. --hidden_size 128 --layer_N 2: the general towers
cfg = default_cfg(["--seed", "0", "--lr", "7e-4", "--critic_lr", "7e-4", "--episode_length", str(T),
                   "--use_recurrent_policy", rec, "--use_valuenorm", "true", "--use_adv_normalize", "true",
                   "--amd_perm_mode", "device", "--log_interval", "1000000"] + extra)
env = make("simple_spread", env_num=N, device=dev)
net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=N)
iters = int(os.environ.get("ORL_ITERS", "600"))
cfg.num_env_steps = N * T * iters
class _A: num_time_steps = 0
tr = PPOAlgorithm(cfg, net.module, agent_num=3, device=dev)
buf = NormalReplayBuffer(cfg, 3, env.observation_space, env.action_space, device=dev)
drv = OnPolicyDriver({"cfg": cfg, "num_agents": 3, "run_dir": None, "envs": env, "device": dev}, tr, buf, _A())
drv.reset_and_buffer_init()
for i in range(iters):
    drv.episode = i
    drv._inner_loop()
    if i % 50 == 0 or i == iters - 1:
        print(i, "episode reward (sum over 25 steps, shared):", round(float(buf.data.rewards[:, :, 0, 0].sum(0).mean()), 2), flush=True)
