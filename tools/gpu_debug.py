import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT, os.path.join(ROOT,"tests")]
import numpy as np, torch
import oracle_lib as ol
from steppingstone_amd.envs import SteppingStoneVecEnv
n=128
g=SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=11, device="cuda:0", return_numpy=True)
o=ol.OracleEnv("walker3d",n,seed=11)
g.reset(); o.reset()
names=[("pos",0,3),("quat",3,7),("vel",7,13),("q",13,34),("qd",34,55),("misc",55,65),("terr",65,185)]
for t in range(4):
    st=o.get_state(); g.set_state(st)
    back=g.get_state().cpu().numpy()
    print("set/get roundtrip err", np.abs(back-st).max())
    a=o.random_actions(t)
    oo,ro,do,io=o.step(a); so=o.get_state()
    og,rg,dg,ig=g.step(a); sg=g.get_state().cpu().numpy()
    err=np.abs(sg-so)
    nd=~do.astype(bool)
    print("t",t,"done",do.sum(),dg.sum(),"max state err (not done)",err[nd].max(),"obs",np.abs(og-oo)[nd].max(),"rew",np.abs(rg-ro)[nd].max())
    for nm,a0,a1 in names: print("   ",nm, err[nd][:,a0:a1].max(), "median", np.median(err[nd][:,a0:a1].max(axis=1)))
    e0=np.where(nd)[0][0]
    print("   env",e0,"q oracle",so[e0,13:20],"q gpu",sg[e0,13:20])
    print("   env",e0,"vel oracle",so[e0,7:13],"vel gpu",sg[e0,7:13])
    print("   flags", so[e0,64], sg[e0,64])
