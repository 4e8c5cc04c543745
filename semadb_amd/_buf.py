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
