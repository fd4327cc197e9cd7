"""Run-to-run determinism hunt (run on the GPU box): the same injected state and actions stepped REPS times per robot and env count
(1, 3, 33, 700); any bit of obs / rew / done / state / info that differs from the first result is reported.
   python tools/determinism_hunt.py REPS [numpy]     (numpy: the pinned numpy drop-in mode instead of device tensors)"""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as ol
from steppingstone_amd.envs import SteppingStoneVecEnv
reps = int(sys.argv[1])
NUMPY = len(sys.argv) > 2 and sys.argv[2] == "numpy"
bad = 0
for env_id, kind in (("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike")):
    for n in (1, 3, 33, 700):
        o = ol.OracleEnv(kind, n, seed=2); o.set_curriculum(5); o.reset()
        for t in range(12):
            o.step(o.random_actions(t))
        st = o.get_state().astype(np.float32)
        act = o.random_actions(50) if NUMPY else torch.as_tensor(o.random_actions(50)).cuda()
        g = SteppingStoneVecEnv(env_id, n, seed=2, device="cuda:0", return_numpy=NUMPY)
        g.update_curriculum(5); g.reset()
        std = st if NUMPY else torch.as_tensor(st).cuda()
        ref = None
        for r in range(reps):
            g.set_state(std)
            ob, rw, dn, _ = g.step(act)
            if NUMPY:
                ob, rw, dn = torch.as_tensor(np.array(ob)).cuda(), torch.as_tensor(np.array(rw, np.float32)).cuda(), torch.as_tensor(np.array(dn)).cuda()
            out = torch.cat([ob.flatten(), rw.flatten(), dn.flatten().float(), g.get_state().flatten(), g._info.flatten().float()])
            if ref is None:
                ref = out.clone()
            elif not torch.equal(out, ref):
                bad += 1
                idx = torch.nonzero(out != ref).flatten()[:6].tolist()
                print("DIFF", env_id, n, "rep", r, "idx", idx, [float(out[i]) for i in idx], [float(ref[i]) for i in idx], flush=True)
        g.close()
        print(env_id, n, "done", flush=True)
print("nondeterministic results:", bad)
