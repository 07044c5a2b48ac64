"""On-disk cache of PACKED weights (checkpoint compatibility + fold/pack once + cache).

Turning a checkpoint of the reference (train/train.py:120-128: ``{'model_state_dict': ..., ...}``) into what the HIP
path consumes costs host work per layer: eval-BN folding in f64, HarDNet's per-link weight gathering, OIHW -> [Cout][KH][KW][Cin]
packing, the fused RPN / head weight stacks.  The result depends only on the state_dict, so it is computed once and
kept on disk:

    <dir>/<backbone>-<sha256 of state_dict + config, 32 hex>.tsodpack torch.save of {"format", "backbone", "hash",
                                                                     "entries": {owner: {layer key: packed state}}}
    <dir>/tiles-<backbone>-<N>x<H>x<W>-<device name>.json             a bare (tile, split_k) list per input geometry (save_tiles)
    <dir>/tuning-<backbone>-<N>x<H>x<W>-<key, 32 hex>.json            what FasterRCNN.tune returned (tables of both schedules, head
                                                                     GEMM choices, launch structure, demotions) - keyed by the
                                                                     weights + config hash above, the device name, the input
                                                                     geometry, the tuning arguments and the sha256 of libtsod.so

The packed state of a layer is plain data (tensors, numbers, strings, lists): the file loads with ``weights_only=True``.
Everything in it is keyed by ONE hash over what the packed form depends on: the state_dict (keys, dtypes, shapes, bytes - so
the layer set too) AND the inputs that are not in the state_dict (``config_fingerprint``: base anchors, feat_stride, RoI op,
BatchNorm eps values, shortcut fusion, this file's FORMAT).  A stale file can therefore never be applied to other weights or to
a detector with other anchors; a file whose format / backbone / owner names do not match is ignored and rewritten.
"""
from __future__ import annotations

import hashlib
import json
import os

import torch

from ._ffi import TsodError

FORMAT = 4


# ----------------------------------------------------------------------------- hashing
def state_dict_hash(sd) -> str:
    """sha256 over (key, dtype, shape, bytes) of every entry in key order (32 hex digits kept)."""
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        v = sd[k]
        t = v.detach().cpu().contiguous() if isinstance(v, torch.Tensor) else torch.as_tensor(v)
        h.update(k.encode())
        h.update(str(t.dtype).encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(t.reshape(-1).view(torch.uint8).numpy().tobytes() if t.numel() else b"")
    return h.hexdigest()[:32]


def config_fingerprint(model) -> bytes:
    """What the packed entries depend on besides the state_dict: the RPN entry stores the base anchors (ratios /
    anchor_scales), the folded weights depend on every BatchNorm's eps, the plan on shortcut fusion and the RoI op."""
    import numpy as np
    h = hashlib.sha256()
    h.update(f"format{FORMAT}|stride{model.feat_stride}|roi{model.head.roi_op}|fuse{int(bool(model.extractor.fuse_shortcut))}|".encode())
    base = model.rpn.anchor_base
    base = base.detach().cpu().numpy() if isinstance(base, torch.Tensor) else np.asarray(base)
    h.update(np.ascontiguousarray(base.astype(np.float32)).tobytes())
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            h.update(f"{name}:{m.eps!r}:{int(m.affine)}|".encode())
    return h.digest()


# ----------------------------------------------------------------------------- (de)serialising packed objects
def _to_state(obj):
    """Packed-cache value -> plain data.  Objects (engine.PackedConv, models.hardnet._RawConv) become
    {"__obj__": class name, attr: value ...}; tensors go to the CPU; tuples become {"__tuple__": [...]}."""
    if obj is None or isinstance(obj, (int, float, str, bool)):
        return obj
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, (tuple, list)):
        return {"__tuple__": [_to_state(o) for o in obj]}
    name = type(obj).__name__
    if name in ("PackedConv", "_RawConv", "FusedShortcutConv", "FusedBottleneckWeights", "FusedStemWeights"):
        return {"__obj__": name, **{k: _to_state(v) for k, v in vars(obj).items()}}
    raise TsodError(f"weight cache: cannot serialise a {name}")


def _from_state(st, device):
    if st is None or isinstance(st, (int, float, str, bool)):
        return st
    if isinstance(st, torch.Tensor):
        return st.to(device)
    if isinstance(st, dict) and "__tuple__" in st:
        return tuple(_from_state(o, device) for o in st["__tuple__"])
    if isinstance(st, dict) and "__obj__" in st:
        from .engine import FusedBottleneckWeights, FusedShortcutConv, FusedStemWeights, PackedConv
        from .models.hardnet import _RawConv
        cls = {"PackedConv": PackedConv, "_RawConv": _RawConv, "FusedShortcutConv": FusedShortcutConv,
               "FusedBottleneckWeights": FusedBottleneckWeights, "FusedStemWeights": FusedStemWeights}[st["__obj__"]]
        obj = cls.__new__(cls)                       # no packing work: the attributes ARE the packed form
        for k, v in st.items():
            if k != "__obj__":
                setattr(obj, k, _from_state(v, device))
        return obj
    raise TsodError("weight cache: unknown entry in the cache file")


def _owners(model):
    return {"extractor": model.extractor, "rpn": model.rpn, "head": model.head}


# ----------------------------------------------------------------------------- public API
def cache_path(directory, model) -> str:
    h = hashlib.sha256(state_dict_hash(model.state_dict()).encode() + config_fingerprint(model)).hexdigest()[:32]
    return os.path.join(directory, f"{model.backbone}-{h}.tsodpack")


def pack_all(model, device) -> None:
    """Fill every packed-weight cache of the detector for ``device`` without running a forward: the backbone's layers are
    visited by building (and dropping) a plan for a minimal geometry."""
    device = torch.device(device)
    if device.type != "cuda":
        raise TsodError("pack_all: a CUDA/ROCm device is required")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if model.training:
        raise TsodError("pack_all: call .eval() first (eval-mode BatchNorm statistics are folded into the weights)")
    with torch.inference_mode():
        model.extractor.build_plan(1, 64, 64, device)       # traverses every layer -> _packed_cache; the plan itself is dropped
        model.rpn._pack(device)
        model.head._pack(device)


def save_packed(model, directory, device=None) -> str:
    """Write the packed weights of ``model`` (packing whatever is not packed yet) and return the file's path."""
    device = torch.device(device) if device is not None else next(model.parameters()).device
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    pack_all(model, device)
    entries = {}
    for name, owner in _owners(model).items():
        entries[name] = {repr(k[0]): _to_state(v) for k, v in owner._packed_cache.items() if k[1] == device}
    os.makedirs(directory, exist_ok=True)
    path = cache_path(directory, model)
    tmp = path + f".tmp{os.getpid()}"
    torch.save({"format": FORMAT, "backbone": model.backbone, "hash": os.path.basename(path).split("-")[-1].split(".")[0],
                "entries": entries}, tmp)
    os.replace(tmp, path)                                   # atomic: concurrent ranks may race to write the same content
    return path


def load_packed(model, directory, device=None) -> bool:
    """Fill the packed-weight caches of ``model`` from ``directory`` if it holds a file made from exactly these weights.
    Returns False (and leaves the model untouched) when there is none."""
    device = torch.device(device) if device is not None else next(model.parameters()).device
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    path = cache_path(directory, model)
    if not os.path.exists(path):
        return False
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if blob.get("format") != FORMAT or blob.get("backbone") != model.backbone or blob.get("hash") not in path:
        return False
    owners = _owners(model)
    if set(blob["entries"]) != set(owners):
        return False
    import ast
    for name, owner in owners.items():
        for k, st in blob["entries"][name].items():
            owner._packed_cache[(ast.literal_eval(k), device)] = _from_state(st, device)
    return True


def ensure_packed(model, directory, device=None) -> str:
    """load_packed, else pack + save_packed.  Returns "hit" or "miss"."""
    if load_packed(model, directory, device):
        return "hit"
    save_packed(model, directory, device)
    return "miss"


def tiles_path(directory, model, shape, device) -> str:
    name = torch.cuda.get_device_name(device).replace(" ", "_") if torch.cuda.is_available() else "cpu"
    n, _, h, w = (int(v) for v in shape)
    return os.path.join(directory, f"tiles-{model.backbone}-{n}x{h}x{w}-{name}.json")


def save_tiles(model, directory, shape, device, tiles) -> str:
    os.makedirs(directory, exist_ok=True)
    path = tiles_path(directory, model, shape, device)
    tmp = path + f".tmp{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump([list(t) for t in tiles], f)
    os.replace(tmp, path)                                   # atomic: ranks tuning concurrently share the <device name> path
    return path


def load_tiles(model, directory, shape, device):
    path = tiles_path(directory, model, shape, device)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return [tuple(t) for t in json.load(f)]


# ----------------------------------------------------------------------------- the tuned table (FasterRCNN.tune(cache_dir=...))
_LIB_HASH = {}


def library_hash() -> str:
    """sha256 of the libtsod.so this process loaded (16 hex): a table names tile ids and was timed on that library's kernels"""
    from . import _ffi
    path = _ffi.LIB_PATH
    if path not in _LIB_HASH:
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        _LIB_HASH[path] = h.hexdigest()[:16]
    return _LIB_HASH[path]


def tuning_path(directory, model, shape, device, args: dict) -> str:
    """``args``: the JSON-able tuning arguments the table depends on (FasterRCNN.tune passes its own)."""
    name = torch.cuda.get_device_name(device) if torch.cuda.is_available() else "cpu"
    n, _, h, w = (int(v) for v in shape)
    wh = hashlib.sha256(state_dict_hash(model.state_dict()).encode() + config_fingerprint(model)).hexdigest()[:32]
    key = hashlib.sha256(json.dumps({"weights": wh, "device": name, "shape": [n, h, w], "lib": library_hash(), "args": args},
                                    sort_keys=True).encode()).hexdigest()[:32]
    return os.path.join(directory, f"tuning-{model.backbone}-{n}x{h}x{w}-{key}.json")


def save_tuning(model, directory, shape, device, args: dict, table: dict) -> str:
    os.makedirs(directory, exist_ok=True)
    path = tuning_path(directory, model, shape, device, args)
    tmp = path + f".tmp{os.getpid()}"
    with open(tmp, "w") as f:
        json.dump(table, f)
    os.replace(tmp, path)                                   # atomic: ranks tuning concurrently write the same key
    return path


def load_tuning(model, directory, shape, device, args: dict):
    path = tuning_path(directory, model, shape, device, args)
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    return table if isinstance(table, dict) and ("serial" in table or "in_flight" in table) else None
