"""Diagnostic for a schedule-dependent result (run on the GPU box, normally with STEPPINGSTONE_LIB=var/libss_fuzz.so):
    python tools/fuzz_diag.py ENV_ID N HELPERS [STEPS]
Trajectory A: reset + STEPS single-step launches (on-device actions), every step's [packed | info | state] kept.  Then every step of A
is REPLAYED 8 times from A's own state before it (ss_set_state -> one step) and compared with A's record bit for bit; the first few
differing (step, env, column) triples are printed with both bit patterns.  Columns: 0..59 obs | 60 rew | 61 done | 62..67 info |
68..253 state after the step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv

env_id, n, helpers = sys.argv[1], int(sys.argv[2]), sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
os.environ["SS_HELPERS"] = helpers
e = SteppingStoneVecEnv(env_id, n, seed=21, device="cuda:0", return_numpy=False)
e.update_curriculum(5)
e.reset()
packed = torch.zeros((n, 62), device=e.device)


def one(t):
    e.step_packed(packed, actions=None, t=t)
    return torch.cat([packed.view(torch.int32), e._info, e.get_state().view(torch.int32)], dim=1).clone()


states = [e.get_state().clone()]
rec = []
for t in range(steps):
    rec.append(one(t))
    states.append(e.get_state().clone())
nbad = 0
for t in range(steps):
    for rep in range(8):
        e.set_state(states[t])
        r = one(t)
        if not torch.equal(r, rec[t]):
            nbad += 1
            if nbad <= 12:
                idx = torch.nonzero(r != rec[t])
                print("step %d replay %d: %d words differ; first (env, column, replay, recorded): %s" % (
                    t, rep, idx.shape[0], [(i, c, hex(r[i, c].item() & 0xffffffff), hex(rec[t][i, c].item() & 0xffffffff)) for i, c in idx[:10].tolist()]), flush=True)
print("%s n=%d helpers=%s: %d of %d replayed steps differ from the recorded trajectory" % (env_id, n, helpers, nbad, 8 * steps))
