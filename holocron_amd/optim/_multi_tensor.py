"""Chunk tables for the multi-tensor optimizer kernels (hc_mt_chunk in include/holocron_hip.h) and
stream-ordered host->device staging that is safe when the host runs ahead of the GPU."""
import ctypes as C

import numpy as np
import torch

from .._lib import HC_MT_CHUNK, MtChunk

_CHUNK_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("s", "<u8"), ("smax", "<u8"),
                      ("n", "<i4"), ("group", "<i4"), ("tensor", "<i4"), ("flags", "<i4")])
assert _CHUNK_DT.itemsize == C.sizeof(MtChunk)


def chunk_rows(entries):
    """entries: list of dict(p, g, m, s, smax, group, tensor, flags) of fp32 tensors (m/s/smax optional).
    Returns the hc_mt_chunk array as raw uint8 numpy bytes and the chunk count."""
    rows = []
    for e in entries:
        n = e["p"].numel()
        ptrs = [0 if e.get(k) is None else e[k].data_ptr() for k in ("p", "g", "m", "s", "smax")]
        for off in range(0, n, HC_MT_CHUNK):
            cnt = min(HC_MT_CHUNK, n - off)
            rows.append(tuple((q + 4 * off) if q else 0 for q in ptrs) + (cnt, e["group"], e["tensor"], e.get("flags", 0)))
    arr = np.array(rows, dtype=_CHUNK_DT) if rows else np.zeros((0,), dtype=_CHUNK_DT)
    return arr.view(np.uint8).copy(), len(rows)


class VGroups:
    """Launch groups keyed by (param_group index, step count).  The kernels keep one step counter (bias corrections) per group;
    the reference keeps ``state['step']`` per parameter (e.g. holocron/optim/adabelief.py:120-128), and the two only differ when a
    parameter skipped iterations (``grad is None``: conditional branches, layers unfrozen mid-run).  Such parameters simply land in
    a launch group of their own with the same hyper-parameters."""

    def __init__(self):
        self.keys = []
        self._of = {}

    def index(self, gi, step):
        k = (gi, int(step))
        v = self._of.get(k)
        if v is None:
            v = self._of[k] = len(self.keys)
            self.keys.append(k)
        return v

    def __len__(self):
        return len(self.keys)


def build_chunks(entries):
    raw, n = chunk_rows(entries)
    return torch.from_numpy(raw), n


class Staging:
    """A device buffer fed from a small ring of pinned host buffers with async copies.  Every slot
    carries an event so that the host never rewrites a slot whose copy the GPU has not executed
    yet.  Nothing is allocated after construction, so ``upload`` may run inside a hipGraph capture
    (the captured copy node keeps reading its slot; the ring is not reused by the host afterwards
    unless ``upload`` is called again, which then picks another slot)."""

    def __init__(self, nbytes, device, slots=4):
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.host = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.locked = [False] * slots   # slots a captured graph keeps reading on every replay
        self.cur = 0

    def upload(self, raw):
        capturing = torch.cuda.is_current_stream_capturing()
        for _ in range(len(self.host)):
            i = self.cur
            self.cur = (self.cur + 1) % len(self.host)
            if not self.locked[i]:
                break
        else:
            raise RuntimeError("all staging slots are owned by captured graphs")
        if capturing:
            self.locked[i] = True
        if self.events[i] is not None and not capturing:
            self.events[i].synchronize()
        self.host[i].numpy()[:raw.size] = raw
        self.dev.copy_(self.host[i], non_blocking=True)
        if not capturing:
            ev = torch.cuda.Event()
            ev.record()
            self.events[i] = ev
        return self.dev


class DeviceTables:
    """Device-resident copies of an optimizer's small host tables (chunk table, group block, per-tensor arrays), keyed by name and
    re-uploaded only when their BYTES change.  The chunk table of a model is a pure function of the parameter / gradient / state
    addresses, which a training loop keeps from step to step, so after the first step nothing crosses PCIe for it - and a step
    whose tables did not change records no copy node under hipGraph capture.  Uploads go through pinned staging rings (async,
    no host synchronisation)."""

    def __init__(self):
        self.slots = {}

    def get(self, name, raw, device):
        raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        key = raw.tobytes()
        slot = self.slots.get(name)
        if slot is not None and slot[0] == key and slot[1].dev.device == device:
            return slot[1].dev
        n = max(raw.size, 16)
        stage = slot[1] if (slot is not None and slot[1].dev.numel() == n and slot[1].dev.device == device) else Staging(n, device)
        stage.upload(raw)
        self.slots[name] = (key, stage)
        return stage.dev
