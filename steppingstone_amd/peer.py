"""Peer-store all-gather: the per-step exchange of the packed [N/G,62] block
WITHOUT a collective in the data path.  The step kernel of every rank stores its rows straight into every peer's gather
buffer over xGMI (ss_step_packed_peers) and the last workgroup publishes the step number in the peers' flag words;
consumers wait on their own flag array (ss_peer_wait, one wavefront).  Buffers live in fine-grained device memory
(hipExtMallocWithFlags) and cross process boundaries as 64-byte HIP IPC handles.  The RCCL all-gather of
steppingstone_amd.distributed.ShardedVecEnv stays the default; this path is selected with SS_PEER_GATHER=1 /
ShardedVecEnv(..., peer_gather=True) and needs all ranks on one node.

Two ways to connect:
  * PeerGather.connect_processes(local_env): one process per GPU, handles exchanged through torch.distributed
    (all_gather_object -- control plane only, any backend);
  * PeerGather.connect_in_process([env0, env1, ...]): several env handles of ONE process act as the ranks
    ("self-peering": the whole protocol on a single GPU, tests/test_gpu_peer.py).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import OBS_DIM

PACK = OBS_DIM + 2
RING = 2            # gather buffers used in turn: a rank may run one step ahead of its slowest consumer


class _DevMem:
    """A raw device allocation viewed as a torch tensor (no copy) through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.ptr = int(ptr)
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (self.ptr, False), "version": 2}


def _alloc(nbytes):
    lib = _lib.load()
    p = C.c_void_p()
    _lib.check(lib.ss_peer_alloc(C.byref(p), int(nbytes)))
    return p.value


def _handle(ptr):
    buf = (C.c_char * 64)()
    _lib.check(_lib.load().ss_peer_ipc_handle(C.c_void_p(ptr), buf))
    return bytes(buf)


def _open(handle):
    p = C.c_void_p()
    _lib.check(_lib.load().ss_peer_ipc_open(C.c_char_p(handle), C.byref(p)))
    return p.value


class PeerGather:
    def __init__(self, env, rank, world):
        self.env, self.rank, self.world = env, int(rank), int(world)
        self.n_local = int(env.num_envs)
        self.device = torch.device(env.device)
        n_all = self.n_local * self.world
        with torch.cuda.device(self.device):
            self.gather_ptr = [_alloc(n_all * PACK * 4) for _ in range(RING)]
            self.flag_ptr = [_alloc(self.world * 4) for _ in range(RING)]
        self._owners = [_DevMem(p, (n_all, PACK), "<f4") for p in self.gather_ptr]
        self.gathered = [torch.as_tensor(o, device=self.device) for o in self._owners]
        self._opened = []
        self.step_id = 0

    # -- wiring
    def _connect(self, gather_ptrs, flag_ptrs):
        """gather_ptrs[slot][rank], flag_ptrs[slot][rank]: pointers valid in THIS process."""
        g = (C.c_void_p * (RING * self.world))(*[p for row in gather_ptrs for p in row])
        f = (C.c_void_p * (RING * self.world))(*[p for row in flag_ptrs for p in row])
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().ss_peer_connect(self.env.backend.h, self.world, self.rank, RING, g, f))

    @classmethod
    def connect_in_process(cls, envs):
        world = len(envs)
        peers = [cls(e, r, world) for r, e in enumerate(envs)]
        for p in peers:
            p._connect([[q.gather_ptr[s] for q in peers] for s in range(RING)],
                       [[q.flag_ptr[s] for q in peers] for s in range(RING)])
        return peers

    @classmethod
    def connect_processes(cls, env, group=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        me = cls(env, rank, world)
        mine = {"gather": [_handle(p) for p in me.gather_ptr], "flag": [_handle(p) for p in me.flag_ptr]}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        gp, fp = [], []
        with torch.cuda.device(me.device):
            for s in range(RING):
                grow, frow = [], []
                for r in range(world):
                    if r == rank:
                        grow.append(me.gather_ptr[s]); frow.append(me.flag_ptr[s])
                    else:
                        g, f = _open(everyone[r]["gather"][s]), _open(everyone[r]["flag"][s])
                        me._opened += [g, f]
                        grow.append(g); frow.append(f)
                gp.append(grow); fp.append(frow)
        me._connect(gp, fp)
        dist.barrier(group)
        return me

    # -- data path
    def step(self, actions=None, t=0, info=None):
        """One control step whose packed rows land in every peer's gather buffer; returns the slot used."""
        env = self.env
        self.step_id += 1
        slot = self.step_id % RING
        if actions is not None:
            env._act.copy_(actions.reshape(env.num_envs, -1))
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        inf = env._info if info is None else info
        _lib.check(_lib.load().ss_step_packed_peers(env.backend.h, C.c_void_p(env._act.data_ptr()) if actions is not None else None,
                                                   0 if actions is not None else 1, int(t), slot, self.step_id, None,
                                                   C.c_void_p(inf.data_ptr()), stream))
        return slot

    def wait(self, slot):
        """Stream-ordered: work enqueued after this call sees all peers' rows of the current step in gathered[slot]."""
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().ss_peer_wait(self.env.backend.h, slot, self.step_id, stream))
        return self.gathered[slot]

    def error(self):
        v = C.c_uint32(0)
        _lib.check(_lib.load().ss_peer_error(self.env.backend.h, C.byref(v)))
        return int(v.value)

    def close(self):
        lib = _lib.load()
        torch.cuda.synchronize(self.device)
        for p in self._opened:
            lib.ss_peer_ipc_close(C.c_void_p(p))
        self._opened = []
        self.gathered = []
        for p in self.gather_ptr + self.flag_ptr:
            lib.ss_peer_free(C.c_void_p(p))
        self.gather_ptr, self.flag_ptr = [], []
