"""Self-launch of the one-process-per-GPU layout.

The reference fans its workers out itself: `make_vec_envs` builds one thunk per env and `ShmemVecEnv.__init__` forks
one process per thunk (common/envs_utils.py:48-56, 519-538), so `python -m playground.train` needs no external
launcher.  The counterpart here is one process per GPU: when a script is asked for N > 1 GPUs and was NOT started by
`torch.distributed.run` (no WORLD_SIZE in the environment), `ensure_ranks` re-executes it under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>`
with the same argument list and returns the children's exit code.  Started by the launcher already (WORLD_SIZE set),
it only validates the rank count.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched():
    """True inside a rank started by torch.distributed.run / torchrun."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def launcher_command(n, script_argv, port=None, module=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())]
    if module:
        cmd += ["-m", module]
    return cmd + list(script_argv)


def ensure_ranks(n, script_argv=None, module=None):
    """Returns None when this process should carry on as a rank (or as the single process of an N=1 run).
    Otherwise it has run the N ranks as children and returns their exit code: the caller exits with it."""
    n = int(n)
    if launched():
        world = int(os.environ["WORLD_SIZE"])
        if n > 1 and world != n:
            raise SystemExit("asked for %d GPUs but the launcher started %d ranks (WORLD_SIZE)" % (n, world))
        return None
    if n <= 1:
        return None
    argv = list(sys.argv if script_argv is None else script_argv)
    if module:
        argv = argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env.setdefault("OMP_NUM_THREADS", "8")
    # free_port() releases the port before the launcher binds it: if somebody else takes it in between, the rendezvous fails
    # within seconds -- one more attempt with a fresh port then (a run that fails later is not repeated)
    import time
    for attempt in (0, 1):
        t0 = time.time()
        rc = subprocess.call(launcher_command(n, argv, module=module), env=env)
        if rc == 0 or attempt == 1 or time.time() - t0 > 15.0:
            return rc
        print("steppingstone_amd.launch: the ranks exited with %d after %.1f s; retrying once with another port" % (rc, time.time() - t0),
              file=sys.stderr, flush=True)
    return rc
