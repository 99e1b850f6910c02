import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
from helpers import make_pair
from oracle import torch_ref
B,S,hid=256,58,(1024,1024,1024,1024)
dqn, orc, data, rng = make_pair(pkg, B=B, S=S, hidden=hid, n_replay=2048, wscale=5.0)
t = torch_ref.TorchRef(B=B,S=S,hidden=hid)
for net in range(4): t.set_params(net, orc.get_params(net))
s,a,r,mc,nx,term = data
for it in range(2):
    idx = rng.integers(0, 2048, size=B)
    l1,q1 = dqn.UpdateActorCritic(idx); l2,q2 = orc.update(idx)
    l3,q3 = t.update(s[idx],a[idx],r[idx],mc[idx],nx[idx],term[idx])
    print('it',it,'loss',l1,l2,l3,'avgq',q1,q2,q3)
    for k in ('q_target','y','q_train','q_policy','actor_out','dq_da'):
        h,o,f = dqn.debug_read(k), orc.debug_read(k), t.dbg[k].numpy().reshape(dqn.debug_read(k).shape)
        print('  %-10s hip-orc %.3e  hip-f64 %.3e  orc-f64 %.3e  refmax %.3e'%(k,np.abs(h-o).max(),np.abs(h-f).max(),np.abs(o-f).max(),np.abs(f).max()))
    h,o,f = dqn.debug_read('dq_da'), orc.debug_read('dq_da'), t.dbg['dq_da'].numpy()
    bad_ho = np.unique(np.where(np.abs(h-o)>1e-5+2e-3*np.abs(o))[0]); bad_hf=np.unique(np.where(np.abs(h-f)>1e-5+2e-3*np.abs(f))[0]); bad_of=np.unique(np.where(np.abs(o-f)>1e-5+2e-3*np.abs(f))[0])
    print('  rows bad hip-orc',bad_ho.tolist()); print('  rows bad hip-f64',bad_hf.tolist()); print('  rows bad orc-f64',bad_of.tolist())
    for net in range(4):
        print('   w net',net,'hip-orc',np.abs(dqn.get_params(net)-orc.get_params(net)).max(),'hip-f64',np.abs(dqn.get_params(net)-t.get_params(net)).max())
