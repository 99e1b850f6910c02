import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from helpers import make_pair
dqn, orc, data, rng = make_pair(pkg, B=32, S=59, hidden=(1024,512,256,128), n_replay=2048, wscale=5.0)
for it in range(3):
    idx = rng.integers(0, 2048, size=32)
    l1,q1 = dqn.UpdateActorCritic(idx); l2,q2 = orc.update(idx)
    print('it',it,'loss',l1,l2,'avgq',q1,q2)
    for k in ('q_target','y','q_train','q_policy','actor_out','dq_da'):
        a,b = dqn.debug_read(k), orc.debug_read(k)
        print('  ',k,'maxabs',np.abs(a-b).max(),'ref max',np.abs(b).max())
    for kind in (0,3,1,2):
        for net in range(4 if kind==0 else 2):
            a,b = dqn.get_params(net,kind), orc.get_params(net,kind)
            d=np.abs(a-b); i=d.argmax()
            print('   kind',kind,'net',net,'maxdiff',d.max(),'at',i,'of',a.size,'vals',a[i],b[i], 'n>1e-5', (d>1e-5).sum())
