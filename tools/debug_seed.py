import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import mmdfn_oracle as O
import test_edge_cases_gpu as T
from mm_dfn_amd import synthetic
from util import rel_err
seed = 45
rs = np.random.RandomState(900 + seed)
B = int(rs.randint(1, 7)); lengths = [int(x) for x in rs.randint(1, 41, size=B)]
cfg = dict(B=B, L=max(lengths), P=int(rs.randint(2, 10)), C=int(rs.choice([6, 7])), nlayers=int(rs.randint(1, 5)),
           D_t=4 * int(rs.randint(5, 160)), D_a=4 * int(rs.randint(5, 100)), D_v=4 * int(rs.randint(5, 140)))
m, logp, p32a, want = T._run_model(cfg, lengths, 950 + seed)
sd = synthetic.seeded_state_dict(m.state_dict(), 950 + seed)
b = synthetic.make_batch(950 + seed + 1, lengths=lengths, **cfg)
def orc(dt, engine):
    params = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()}
    out = O.forward(params, b["textf"].to(dt), b["qmask"].to(dt), b["umask"].to(dt), b["lengths"], b["acouf"].to(dt), b["visuf"].to(dt), O.default_cfg(cfg["nlayers"]), engine=engine)
    w = torch.from_numpy(np.random.RandomState(950 + seed).randn(*out.shape).astype(np.float32)).to(dt)
    (out * w).sum().backward()
    return {k: v.grad for k, v in params.items()}
g32m, g64m = orc(torch.float32, "manual"), orc(torch.float64, "manual")
named = dict(m.named_parameters())
for k in ("graph_model.graph_net.convs.0.weight", "linear_v.weight"):
    d = named[k].grad
    print(k)
    print("  dev vs o32aten  %.3e" % rel_err(d, p32a[k].grad))
    print("  dev vs o32man   %.3e" % rel_err(d, g32m[k]))
    print("  dev vs o64man   %.3e" % rel_err(d, g64m[k]))
    print("  o32aten vs o64  %.3e" % rel_err(p32a[k].grad, g64m[k]))
    print("  o32man vs o64   %.3e" % rel_err(g32m[k], g64m[k]))
    print("  o32aten vs o32man %.3e" % rel_err(p32a[k].grad, g32m[k]))
