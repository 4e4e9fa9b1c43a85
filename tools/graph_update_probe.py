"""TIMING probe (results are not meaningful training): the 10-epoch PPO update of configuration 2 as plain launches vs one
hipGraph replay of the same launches (by-value arguments - Adam step, seeds - frozen at capture time).  Prices what a
device-side step counter + graph capture of the update would buy.  Usage: graph_update_probe.py [envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.normal_buffer import NormalReplayBuffer
    from openrl_amd.configs.config import default_cfg
    from openrl_amd.drivers.onpolicy_driver import OnPolicyDriver
    from openrl_amd.envs.common import make
    from openrl_amd.modules.common import PPONet

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda:0")
    cfg = default_cfg(["--episode_length", "128", "--ppo_epoch", "10", "--amd_perm_mode", "device", "--log_interval",
                       "1000000"])
    env = make("SyntheticFixedStep-v0", env_num=n, obs_dim=4, episode_limit=200, device=dev, seed=1)
    net = PPONet(env, cfg=cfg, device=dev, n_rollout_threads=n)
    cfg.num_env_steps = n * 128 * 64
    algo = PPOAlgorithm(cfg, net.module, agent_num=1, device=dev)
    buf = NormalReplayBuffer(cfg, 1, env.observation_space, env.action_space, device=dev)

    class _A:
        num_time_steps = 0

    drv = OnPolicyDriver({"cfg": cfg, "num_agents": 1, "run_dir": None, "envs": env, "device": dev}, algo, buf, _A(),
                         rank=0, world_size=1)
    drv.reset_and_buffer_init()
    for i in range(3):
        drv.episode = i
        drv._inner_loop()
    torch.cuda.synchronize()
    data = buf.data
    M = 128 * n
    algo._advantages_and_records(data)
    algo._full_batch_moments = True
    algo._moments = algo._adv_stats[8:11]
    algo._info = torch.zeros(8, device=dev)
    algo._info_first = False

    def epochs():
        nxt = None
        for e in range(10):
            batches, mbs = algo._minibatch_indices(M, nxt)
            nxt = algo._update_minibatch(data, batches[0], mbs, True, algo._perm_job(M) if e < 9 else None)

    def timeit(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    eager = timeit(epochs)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        epochs()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        epochs()
    graph = timeit(g.replay)
    print("envs %d: 10 epochs as plain launches %.4f ms, as one hipGraph replay %.4f ms" % (n, eager, graph))


if __name__ == "__main__":
    main()
