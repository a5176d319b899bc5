"""Chunk tables for the multi-tensor optimizer kernels (hc_mt_chunk in include/holocron_hip.h)."""
import ctypes as C

import numpy as np
import torch

from .._lib import HC_MT_CHUNK, MtChunk

_CHUNK_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("s", "<u8"), ("smax", "<u8"),
                      ("n", "<i4"), ("group", "<i4"), ("tensor", "<i4"), ("flags", "<i4")])
assert _CHUNK_DT.itemsize == C.sizeof(MtChunk)


def build_chunks(entries):
    """entries: list of dict(p, g, m, s, smax, group, tensor, flags) of fp32 tensors (m/s/smax optional).
    Returns a uint8 CPU tensor holding the hc_mt_chunk array and the chunk count."""
    rows = []
    for e in entries:
        n = e["p"].numel()
        ptrs = [0 if e.get(k) is None else e[k].data_ptr() for k in ("p", "g", "m", "s", "smax")]
        for off in range(0, n, HC_MT_CHUNK):
            cnt = min(HC_MT_CHUNK, n - off)
            rows.append(tuple((q + 4 * off) if q else 0 for q in ptrs) + (cnt, e["group"], e["tensor"], e.get("flags", 0)))
    arr = np.array(rows, dtype=_CHUNK_DT) if rows else np.zeros((0,), dtype=_CHUNK_DT)
    return torch.from_numpy(arr.view(np.uint8).copy()), len(rows)
