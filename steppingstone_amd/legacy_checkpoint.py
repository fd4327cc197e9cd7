"""Reader for the reference's shipped / trained checkpoints (playground/models/*.pt, playground/train.py:523-562).

The reference saves the whole `Policy` module with the legacy (pre-zip) `torch.save` format: pickles that name the reference's
own classes (`common.controller.Policy`, `SoftsignActor`, ...) followed by raw float32 storages.  `torch.load` of such a file
needs those classes importable and executes whatever the pickle says.  This reader does neither: a restricted unpickler that
maps every class to an inert stub and every tensor to (storage key, offset, size, stride), then rebuilds the arrays from the raw
storages (layout: SURVEY.md section 10).  `load_reference_checkpoint(path)` returns a `steppingstone_amd.ppo.ActorCritic` with
the file's actor, log-std and critic weights -- the loader ADVICE r2 asked for, since this package's own checkpoints are
state_dicts (`ppo.save_checkpoint`) and the reference's `enjoy.py`-style `torch.load` cannot read those.
"""
import collections
import pickle
import struct

import numpy as np


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {})


def _rebuild_tensor_v2(storage, offset, size, stride, *rest):
    return ("tensor", storage, offset, tuple(size), tuple(stride))


def _rebuild_parameter(data, requires_grad, hooks):
    return data


class _U(pickle.Unpickler):
    def find_class(self, mod, name):
        if (mod, name) == ("collections", "OrderedDict"):
            return collections.OrderedDict
        if (mod, name) == ("torch._utils", "_rebuild_tensor_v2"):
            return _rebuild_tensor_v2
        if (mod, name) == ("torch._utils", "_rebuild_parameter"):
            return _rebuild_parameter
        if mod == "torch" and name.endswith("Storage"):
            return name
        return _Stub          # model classes / backends -> inert stubs

    def persistent_load(self, pid):
        if pid[0] == "module":
            return pid[1]
        if pid[0] == "storage":
            return ("storage", pid[2], pid[4])        # key, numel
        raise pickle.UnpicklingError(pid)


def read_legacy(path):
    f = open(path, "rb")
    for _ in range(3):
        pickle.load(f)                                # magic, protocol, sys info
    obj = _U(f).load()
    keys = pickle.load(f)
    storages = {}
    for k in keys:
        n = struct.unpack("<q", f.read(8))[0]
        storages[k] = np.frombuffer(f.read(4 * n), dtype="<f4").copy()
    return obj, storages


def tensors_of(obj, storages, prefix="", out=None):
    out = {} if out is None else out
    d = getattr(obj, "__dict__", {})
    for group in ("_parameters", "_buffers"):
        for k, v in (d.get(group) or {}).items():
            if isinstance(v, tuple) and v and v[0] == "tensor":
                _, st, off, size, stride = v
                flat = storages[st[1]]
                out[prefix + k] = np.lib.stride_tricks.as_strided(flat[off:], size, [s * 4 for s in stride]).copy()
    for k, m in (d.get("_modules") or {}).items():
        tensors_of(m, storages, prefix + k + ".", out)
    return out




def load_reference_checkpoint(path, device="cpu"):
    """ActorCritic (one critic) carrying the weights of a reference checkpoint: actor.fc1..fc5 / out, dist.logstd._bias ->
    logstd, critic.{0,2,4,6,8} -> critics.0.*.  Raises KeyError if the file does not hold that architecture."""
    import torch
    from . import ppo
    obj, storages = read_legacy(path)
    w = tensors_of(obj, storages)
    ac = ppo.ActorCritic(num_ensembles=1)
    sd = {}
    for k, v in w.items():
        if k.startswith("actor."):
            sd[k] = torch.from_numpy(v)
        elif k == "dist.logstd._bias":
            sd["logstd"] = torch.from_numpy(v.reshape(-1))
        elif k.startswith("critic."):
            sd["critics.0." + k[len("critic."):]] = torch.from_numpy(v)
    missing = set(ac.state_dict()) - set(sd)
    if missing:
        raise KeyError("not a SoftsignActor / critic checkpoint: %s lacks %s" % (path, sorted(missing)))
    ac.load_state_dict(sd)
    return ac.to(device)
