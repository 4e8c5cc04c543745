"""Buffer helpers: numpy arrays are host memory, torch CUDA tensors are HBM."""
import ctypes as C

import numpy as np

from ._lib import MEM_DEVICE, MEM_HOST


def is_torch_cuda(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def as_f32(x):
    """Returns (obj_kept_alive, pointer, mem, shape)."""
    if is_torch_cuda(x):
        import torch
        t = x.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        return t, C.c_void_p(t.data_ptr()), MEM_DEVICE, tuple(t.shape)
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a, C.c_void_p(a.ctypes.data), MEM_HOST, a.shape


def np_ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def current_stream(mem):
    if mem != MEM_DEVICE:
        return None
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def empty_like_mem(mem, shape, dtype, device_index=0, zero=False):
    """dtype: numpy dtype name ('float32', 'uint64', 'uint32', 'uint8').  Device buffers are NOT cleared unless
    `zero`: every kernel that fills them writes every element (short result rows are zero-padded by the kernel),
    and a fill kernel per output would cost more launch time than it is worth next to a 1 ms search."""
    if mem == MEM_DEVICE:
        import torch
        tdt = {"float32": torch.float32, "uint64": torch.int64, "uint32": torch.int32, "uint8": torch.uint8}[dtype]
        t = (torch.zeros if zero else torch.empty)(shape, dtype=tdt, device="cuda:%d" % device_index)
        return t, C.c_void_p(t.data_ptr())
    a = np.zeros(shape, dtype=dtype)
    return a, C.c_void_p(a.ctypes.data)


class _PinnedBlock:
    """owns one sdb_host_alloc block for as long as an array made over it lives"""

    def __init__(self, nbytes):
        from ._lib import check, lib
        self._p = C.c_void_p(0)
        check(lib().sdb_host_alloc(max(1, int(nbytes)), C.byref(self._p)))
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            from ._lib import lib
            if self._p:
                lib().sdb_host_free(self._p)
                self._p = C.c_void_p(0)
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array over page-locked host memory (sdb_host_alloc): a host-memory search reads / writes it in place"""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    blk = _PinnedBlock(n * dt.itemsize)
    raw = (C.c_char * max(1, n * dt.itemsize)).from_address(blk._p.value)
    arr = np.frombuffer(raw, dtype=dt, count=n).reshape(shape)
    _KEEP[id(raw)] = blk  # numpy keeps `raw` alive through arr.base; the block goes when `raw` does
    import weakref
    weakref.finalize(raw, _KEEP.pop, id(raw), None)
    return arr


_KEEP = {}
