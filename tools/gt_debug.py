import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.test_gen_tower_gpu import make_net, torch_reference, DEV
from openrl_amd import spaces
for (H, layer_N, act_id, fn, D, B) in [(128, 1, 1, False, 4, 1000), (64, 1, 1, False, 4, 128), (64, 2, 0, True, 17, 300)]:
    cfg, net = make_net("policy", H, layer_N, act_id, fn, D, spaces.Discrete(3))
    ft = net.gt(("act",))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, D, generator=g)
    dh = torch.randn(B, 3, generator=g) / B
    outs, gref = torch_reference(net, ("act",), x.double(), [dh.double()])
    ft.prep()
    net.grad.fill_(float("nan"))
    ft.backward(x.to(DEV), 0, None, B, dh.to(DEV).contiguous())
    torch.cuda.synchronize()
    got = net.grad.cpu().double()
    print("case", H, layer_N, D, B)
    for key, shape, off in net.entries:
        n = int(np.prod(shape))
        a, b = got[off:off+n], gref[off:off+n]
        if ".fc_h." in key or "logstd" in key: continue
        sc = max(b.abs().max().item(), 1e-9)
        print("  %-28s relerr %.3e  finite %s  |ref| %.3e |got| %.3e" % (key, (a-b).abs().max().item()/sc, bool(torch.isfinite(a).all()), sc, a.abs().max().item()))
