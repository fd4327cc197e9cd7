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


_STORAGE_DTYPES = {"FloatStorage": "<f4", "DoubleStorage": "<f8", "HalfStorage": "<f2", "LongStorage": "<i8", "IntStorage": "<i4",
                   "ShortStorage": "<i2", "CharStorage": "i1", "ByteStorage": "u1", "BoolStorage": "?"}
_MAGIC = 0x1950A86A20F9469CFC6C          # torch/serialization.py MAGIC_NUMBER of the legacy (pre-zip) format


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
            stype = pid[1] if isinstance(pid[1], str) else getattr(pid[1], "__name__", str(pid[1]))
            if stype not in _STORAGE_DTYPES:
                raise pickle.UnpicklingError("unknown storage type %r" % (pid[1],))
            self.storage_types[pid[2]] = stype
            return ("storage", pid[2], pid[4], stype)        # key, numel, type
        raise pickle.UnpicklingError(pid)


class _Plain(pickle.Unpickler):
    """The four header / trailer records (magic, protocol, sys-info dict, storage-key list) are plain ints, dicts, strings and
    lists: a record that names ANY global is not one of them and is refused -- nothing in the file is ever executed."""

    def find_class(self, mod, name):
        raise pickle.UnpicklingError("header record of a legacy checkpoint names a global (%s.%s): refused" % (mod, name))

    def persistent_load(self, pid):
        raise pickle.UnpicklingError("header record of a legacy checkpoint holds a persistent id: refused")


def read_legacy(path):
    """(module object of stubs, {storage key: 1-D numpy array in the storage's own dtype})."""
    with open(path, "rb") as f:
        magic = _Plain(f).load()
        if magic != _MAGIC:
            raise ValueError("%s is not a legacy (pre-zip) torch.save file (magic %r)" % (path, magic))
        _Plain(f).load()                              # protocol version
        _Plain(f).load()                              # sys info
        u = _U(f)
        u.storage_types = {}
        obj = u.load()
        keys = _Plain(f).load()
        if not isinstance(keys, (list, tuple)):
            raise ValueError("%s: storage-key record is %r, not a list" % (path, type(keys).__name__))
        storages = {}
        for k in keys:
            n = struct.unpack("<q", f.read(8))[0]
            dt = np.dtype(_STORAGE_DTYPES[u.storage_types.get(k, "FloatStorage")])
            raw = f.read(dt.itemsize * n)
            if len(raw) != dt.itemsize * n:
                raise ValueError("%s: storage %s is truncated" % (path, k))
            storages[k] = np.frombuffer(raw, dtype=dt).copy()
    return obj, storages


def tensors_of(obj, storages, prefix="", out=None):
    out = {} if out is None else out
    d = getattr(obj, "__dict__", {})
    for group in ("_parameters", "_buffers"):
        for k, v in (d.get(group) or {}).items():
            if isinstance(v, tuple) and v and v[0] == "tensor":
                _, st, off, size, stride = v
                flat = storages[st[1]]
                out[prefix + k] = np.lib.stride_tricks.as_strided(flat[off:], size, [s * flat.itemsize for s in stride]).copy()
    for k, m in (d.get("_modules") or {}).items():
        tensors_of(m, storages, prefix + k + ".", out)
    return out


def load_reference_checkpoint(path, device="cpu"):
    """ActorCritic carrying the weights of a reference checkpoint: actor.fc1..fc5 / out, dist.logstd._bias -> logstd, and the value
    networks: the current reference registers its ensemble as modules c0, c1, ... (common/controller.py:94-95; the shipped Mike
    policy) -> critics.{i}.*, older files hold one module `critic` (the shipped Walker3D policies) -> critics.0.*.  The ensemble
    size and the state / action dimensions are taken from the file.  Raises KeyError if the file does not hold that architecture."""
    import re
    import torch
    from . import ppo
    obj, storages = read_legacy(path)
    w = tensors_of(obj, storages)
    sd, ens = {}, set()
    for k, v in w.items():
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
        m = re.match(r"c(\d+)\.(.*)", k)
        if k.startswith("actor."):
            sd[k] = t
        elif k == "dist.logstd._bias":
            sd["logstd"] = t.reshape(-1)
        elif k.startswith("critic."):
            sd["critics.0." + k[len("critic."):]] = t
            ens.add(0)
        elif m:
            sd["critics.%d.%s" % (int(m.group(1)), m.group(2))] = t
            ens.add(int(m.group(1)))
    if "actor.fc1.weight" not in sd or "logstd" not in sd or not ens or ens != set(range(len(ens))):
        raise KeyError("not a SoftsignActor / critic checkpoint: %s holds %s" % (path, sorted(w)[:8]))
    ac = ppo.ActorCritic(state_dim=sd["actor.fc1.weight"].shape[1], action_dim=sd["logstd"].numel(), num_ensembles=len(ens))
    missing = set(ac.state_dict()) - set(sd)
    if missing:
        raise KeyError("not a SoftsignActor / critic checkpoint: %s lacks %s" % (path, sorted(missing)))
    ac.load_state_dict(sd)
    return ac.to(device)
