import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
from helpers import make_pair
from oracle import torch_ref
B,S,hid=256,58,(1024,1024,1024,1024)
dqn, orc, data, rng = make_pair(pkg, B=B, S=S, hidden=hid, n_replay=2048, wscale=2.0)
t = torch_ref.TorchRef(B=B,S=S,hidden=hid)
for net in range(4): t.set_params(net, orc.get_params(net))
s,a,r,mc,nx,term = data
def cmp(name,h,o,f):
    h=h.astype(np.float64);o=o.astype(np.float64);f=np.asarray(f,np.float64)
    n=np.linalg.norm(f)
    print('  %-12s fro hip-orc %.2e hip-f64 %.2e orc-f64 %.2e | max hip-orc %.2e hip-f64 %.2e orc-f64 %.2e scale %.2e'%(name,np.linalg.norm(h-o)/n,np.linalg.norm(h-f)/n,np.linalg.norm(o-f)/n,np.abs(h-o).max(),np.abs(h-f).max(),np.abs(o-f).max(),np.abs(f).max()))
for it in range(3):
    idx = rng.integers(0, 2048, size=B)
    dqn.update_phase(0, idx); orc.update_phase(0, idx)
    gc_h, gc_o = dqn.get_params(1,3), orc.grad_view(1).copy()
    dqn.update_phase(1); orc.update_phase(1, idx)
    ga_h, ga_o = dqn.get_params(0,3), orc.grad_view(0).copy()
    dqn.update_phase(2); orc.update_phase(2, idx)
    t.update(s[idx],a[idx],r[idx],mc[idx],nx[idx],term[idx])
    print('it',it, dqn.read_stats(), orc.last_stats())
    cmp('critic grad', gc_h, gc_o, t.g[1].numpy())
    cmp('actor grad', ga_h, ga_o, t.g[0].numpy())
    for k in ('q_train','q_policy','actor_out','dq_da'):
        cmp(k, dqn.debug_read(k).ravel(), orc.debug_read(k).ravel(), t.dbg[k].numpy().ravel())
