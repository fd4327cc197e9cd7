import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"tests")]
import numpy as np, torch
import oracle_lib as ol
from steppingstone_amd.envs import SteppingStoneVecEnv
n=128
names=[("pos",0,3),("quat",3,7),("vel",7,13),("q",13,34),("qd",34,55),("misc",55,65)]
for lift in (5.0, 0.0):
    g=SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=11, device="cuda:0", return_numpy=True)
    o=ol.OracleEnv("walker3d",n,seed=11)
    g.reset(); o.reset()
    st=o.get_state(); st[:,2]+=lift; st[:,56]+=lift; o.set_state(st)
    for t in range(2):
        st=o.get_state(); g.set_state(st)
        a=o.random_actions(t)*(0.0 if t==0 else 1.0)
        oo,ro,do,io=o.step(a); so=o.get_state()
        og,rg,dg,ig=g.step(a); sg=g.get_state().cpu().numpy()
        err=np.abs(sg-so)
        nd=~(do.astype(bool)|dg)
        print("lift",lift,"t",t,"done",do.sum(),dg.sum(),"max state err (not done)",err[nd].max() if nd.any() else None)
        if nd.any():
            for nm,a0,a1 in names: print("   ",nm, err[nd][:,a0:a1].max(), "median", np.median(err[nd][:,a0:a1].max(axis=1)))
    g.close()
