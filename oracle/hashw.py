"""ORACLE (test infrastructure): synthetic weights from the counter-based generator the device also runs
(vita_amd/csrc/vh_elem.hip hash_bf16 / C ABI vh_fill_hash_bf16), so the fp32 oracle and the bf16 device tensors hold
identical values without a checkpoint and without a host copy of the 94 GB model.  Native twin: oracle/csrc/hashfill.c
(OpenMP); a vectorised numpy fallback computes the same integers when the shared object has not been built."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def tensor_seed(name, base=0):
    """64-bit stream id of a reference-named tensor: FNV-1a over the name, mixed with the run's base seed."""
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (int(base) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def _native():
    global _lib
    if _lib is None and os.path.exists(_SO):
        lib = C.CDLL(_SO)
        lib.hash_fill_f32.restype = None
        lib.hash_fill_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64]
        _lib = lib
    return _lib


def _numpy_values(seed, idx):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    m = np.uint64(63)
    v = ((z & m) + ((z >> np.uint64(6)) & m) + ((z >> np.uint64(12)) & m) + ((z >> np.uint64(18)) & m)).astype(np.int64) - 126
    return v.astype(np.float32) * np.float32(2.0 ** -11)


def fill(shape, seed, out=None, ld_src=None, idx0=0):
    """float32 [rows, cols] (or 1-D) with element (r, c) = sample idx0 + r * ld_src + c of stream `seed`."""
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    rows, cols = (1, shape[0]) if len(shape) == 1 else (int(np.prod(shape[:-1])), shape[-1])
    ld_src = cols if ld_src is None else int(ld_src)
    if out is None:
        out = np.empty(shape, np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == rows * cols
    lib = _native()
    if lib is not None:
        lib.hash_fill_f32(out.ctypes.data, rows, cols, cols, ld_src, int(idx0), C.c_uint64(int(seed)))
    else:
        flat = out.reshape(rows, cols)
        step = max(1, (1 << 22) // max(cols, 1))
        for r0 in range(0, rows, step):
            r = np.arange(r0, min(rows, r0 + step), dtype=np.uint64)[:, None]
            idx = np.uint64(idx0) + r * np.uint64(ld_src) + np.arange(cols, dtype=np.uint64)[None, :]
            flat[r0:r0 + r.shape[0]] = _numpy_values(seed, idx)
    return out
