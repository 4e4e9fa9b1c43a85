"""Worker for tests/test_multirank_gpu.py (launched with torch.distributed.run, 2 - 8 ranks, gloo, one GPU).

"G ranks == 1 rank on the concatenated batch" (SURVEY.md section 8e): a global buffer of 2*Nl env lanes is split
by env between the ranks; every rank runs PPOAlgorithm.train on its shard with the all-reduces of
openrl_amd/distributed.py in the loop; rank 0 also runs the same update single-process on the whole buffer and
compares weights, ValueNorm state and train_info."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(cfg, N, D, n_act, world_size, dev, seed=0):
    from openrl_amd import spaces
    from openrl_amd.algorithms.ppo import PPOAlgorithm
    from openrl_amd.buffers.replay_data import ReplayData
    from openrl_amd.modules.ppo_module import PPOModule

    cfg.n_rollout_threads, cfg.num_agents, cfg.rnn_hidden_size = N, 1, cfg.hidden_size
    obs_space, act_space = spaces.Box(-np.inf, np.inf, (D,)), spaces.Discrete(n_act)
    torch.manual_seed(seed)
    module = PPOModule(cfg, obs_space, obs_space, act_space, device=dev, rank=0, world_size=world_size)
    buf = ReplayData(cfg, 1, obs_space, act_space, device=dev)
    return module, buf, PPOAlgorithm(cfg, module, agent_num=1, device=dev)


def fill(buf, host, lo, hi, nv):
    for k, v in host.items():
        getattr(buf, k).copy_(torch.tensor(v[:, lo:hi]))
    return torch.tensor(nv[lo:hi])


def comm_main(du, rank, world, dev):
    """orl_allreduce_small (one-shot P2P push over hipIpc-mapped peer memory; all ranks share cuda:0 here) against an
    all_gather of the contributions summed explicitly in RANK ORDER in fp32 - the order the kernel promises - so the
    comparison is bit-exact for any world size (torch.distributed's own all_reduce orders the sum differently beyond
    2 ranks); many back-to-back collectives of different sizes exercise the parity double-buffering and the
    [parity][source rank][capacity] stride of the inboxes."""
    comm = du.make_small_allreduce(20000, dev)
    ok = comm is not None
    if ok:
        rs = np.random.RandomState(100 + rank)
        for k, n in enumerate([1, 63, 64, 9702, 20000, 7, 9702, 9702, 4096, 1]):
            x = torch.tensor(rs.randn(n).astype(np.float32) * (1 + k), device=dev)
            parts = [torch.zeros_like(x) for _ in range(world)]
            torch.distributed.all_gather(parts, x)  # gloo on CUDA tensors
            want = torch.zeros_like(x)
            for r in range(world):  # 0 + x_0 + x_1 + ... : comm_sum's order (csrc/orl_comm.h)
                want = want + parts[r]
            got = comm.allreduce_(x.clone())
            torch.cuda.synchronize()
            if not torch.equal(got, want):
                ok = False
                print("rank %d: collective %d (n=%d) differs: max|d| %.3e" % (rank, k, n, (got - want).abs().max().item()))
        comm.check()
        assert int(comm.error_flag().item()) == 0
        # every rank holds the identical vector
        g = [torch.zeros_like(got) for _ in range(world)]
        torch.distributed.all_gather(g, got)
        ok &= all(torch.equal(g[0], t) for t in g)
        comm.close()
    flag = torch.tensor([1.0 if ok else 0.0])
    torch.distributed.all_reduce(flag)
    if rank == 0:
        print("COMM_OK" if flag.item() == world else "COMM_FAIL")
    torch.distributed.destroy_process_group()
    sys.exit(0 if flag.item() == world else 1)


def main():
    from openrl_amd import distributed as du
    from openrl_amd.configs.config import default_cfg

    du.init_from_env(backend="gloo")
    rank, world = du.rank(), du.world_size()
    dev = "cuda:0"
    if "comm" in sys.argv[1:]:
        return comm_main(du, rank, world, dev)
    collective = "rccl" if "rccl" in sys.argv[1:] else "p2p"
    fallback = "fallback" in sys.argv[1:]  # one rank's comm self-test fails -> EVERY rank must fall back, in step
    if fallback:
        du._FAULT_INJECT = {"rank": world - 1, "stage": "selftest"}
    T, Nl, D, n_act = 12, 24, 4, 2
    # "odd": a global env count the ranks do not divide (shard_range hands the remainder to the low ranks)
    N = Nl * world + ((world - 1) if "odd" in sys.argv[1:] else 0)
    rs = np.random.RandomState(5)
    host = dict(policy_obs=rs.randn(T + 1, N, 1, D).astype(np.float32), rewards=rs.rand(T, N, 1, 1).astype(np.float32),
                value_preds=(0.3 * rs.randn(T + 1, N, 1, 1)).astype(np.float32),
                masks=(rs.rand(T + 1, N, 1, 1) > 0.05).astype(np.float32),
                active_masks=(rs.rand(T + 1, N, 1, 1) > 0.1).astype(np.float32),
                actions=rs.randint(0, n_act, (T, N, 1, 1)).astype(np.float32),
                action_log_probs=(np.log(0.5) + 0.05 * rs.randn(T, N, 1, 1)).astype(np.float32))
    nv = (0.3 * rs.randn(N, 1, 1)).astype(np.float32)
    argv = ["--episode_length", str(T), "--ppo_epoch", "3", "--num_mini_batch", "1", "--amd_perm_mode", "device",
            "--amd_collective", collective]
    mode = sys.argv[1] if len(sys.argv) > 1 else "mlp"
    recurrent = mode in ("rnn", "genrnn")
    generic = mode in ("gen", "genrnn")  # general towers: torch.distributed all-reduces of the flat gradients
    Hs = 64
    if generic:
        Hs = 96
        argv += ["--hidden_size", str(Hs), "--layer_N", "2", "--activation_id", "0"]
    if recurrent:  # T is even: chunks of 2 never straddle env lanes, so shards see the same chunks as one process
        argv += ["--use_recurrent_policy", "true", "--data_chunk_length", "2"]
        host["rnn_states"] = (0.3 * rs.randn(T + 1, N, 1, 1, Hs)).astype(np.float32)
        host["rnn_states_critic"] = (0.3 * rs.randn(T + 1, N, 1, 1, Hs)).astype(np.float32)

    # sharded run (all ranks)
    cfg = default_cfg(argv)
    lo, hi = du.shard_range(N, rank, world)
    sizes = [du.shard_range(N, r, world) for r in range(world)]
    assert sizes[0][0] == 0 and sizes[-1][1] == N and all(a[1] == b[0] for a, b in zip(sizes, sizes[1:]))
    Nl = hi - lo
    module, buf, algo = build(cfg, Nl, D, n_act, world, dev)
    buf.compute_returns(fill(buf, host, lo, hi, nv), module.get_critic_value_normalizer())
    assert algo.generic == generic
    # general towers wider than PPOAlgorithm.P2P_MAX_FLOATS send their flat gradient vector through torch.distributed:
    # the one-shot push is a latency-regime collective (round-3 ADVICE: every rank pushes the whole vector to every peer)
    small = (not generic) or algo._gen_flat.numel() <= algo.P2P_MAX_FLOATS
    assert (algo._comm is not None) == (collective == "p2p" and not fallback and small), \
        "the fused orl_comm path must be the one that runs (and must be off on every rank after a failed self-test)"
    if generic and not small:
        print("GEN_FLAT_OVER_P2P_THRESHOLD n=%d -> torch.distributed" % algo._gen_flat.numel(), flush=True)
    if fallback:
        du._FAULT_INJECT = None
    info = algo.train(buf)
    torch.cuda.synchronize()
    if algo._comm is not None:
        algo._comm.check()
    ok = True
    # the other collective on the same shard: the fused one-shot push (orl_comm) and the torch.distributed all-reduce
    # add the same two fp32 vectors -> bit-identical weights for 2 ranks
    other = "rccl" if collective == "p2p" else "p2p"
    cfg2 = default_cfg([a if a != collective else other for a in argv])
    m2, b2, a2 = build(cfg2, Nl, D, n_act, world, dev)
    b2.compute_returns(fill(b2, host, lo, hi, nv), m2.get_critic_value_normalizer())
    a2.train(b2)
    torch.cuda.synchronize()
    # (general towers have one collective only: this second run then checks run-to-run determinism of the sharded update)
    if world == 2 or fallback:  # a + b is commutative; beyond 2 ranks gloo's reduction order is not the rank order
        same_coll = all(torch.equal(module.models[k].theta, m2.models[k].theta) for k in ("policy", "critic"))
    else:
        same_coll = all(torch.allclose(module.models[k].theta, m2.models[k].theta, rtol=1e-4, atol=2e-6)
                        for k in ("policy", "critic"))
    flag = torch.tensor([1.0 if same_coll else 0.0])
    torch.distributed.all_reduce(flag)
    if rank == 0:
        print("COLLECTIVES_BITWISE_EQUAL" if flag.item() == world else "COLLECTIVES_DIFFER")
    ok &= flag.item() == world
    if rank == 0:
        cfg1 = default_cfg(argv)
        m1, b1, a1 = build(cfg1, N, D, n_act, 1, dev)
        b1.compute_returns(fill(b1, host, 0, N, nv), m1.get_critic_value_normalizer())
        info1 = a1.train(b1)
        for k in ("policy", "critic"):
            got, want = module.models[k].theta.cpu().numpy(), m1.models[k].theta.cpu().numpy()
            err = np.abs(got - want).max()
            print("theta %s max|diff| %.3e" % (k, err))
            ok &= bool(np.allclose(got, want, rtol=1e-4, atol=2e-6))
        vs, vs1 = module.get_critic_value_normalizer().state.cpu().numpy(), m1.get_critic_value_normalizer().state.cpu().numpy()
        ok &= bool(np.allclose(vs, vs1, rtol=1e-6))
        for k in info1:
            ok &= bool(np.isclose(info[k], info1[k], rtol=2e-4, atol=2e-6))
        print("info", info, info1)
        print("MULTIRANK_EQUIV_OK" if ok else "MULTIRANK_EQUIV_FAIL")
    # all ranks must end with bit-identical weights (same reduced vector, same Adam step)
    th = module.models["policy"].theta.clone()
    gathered = [torch.zeros_like(th) for _ in range(world)]
    torch.distributed.all_gather(gathered, th)
    if rank == 0:
        same = all(torch.equal(gathered[0], g) for g in gathered)
        print("REPLICAS_IDENTICAL" if same else "REPLICAS_DIVERGED")
    torch.distributed.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
