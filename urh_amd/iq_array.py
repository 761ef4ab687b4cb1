"""Device-side counterpart of the reference's `IQArray` conversions
(/root/reference/src/urh/signalprocessing/IQArray.py): `convert_to` (:127-203), `as_complex64` (:92-93) and
`from_file` (:205-227) for captures that live in HBM.  Tensors are (N, 2) of int8 / uint8 / int16 / uint16 / float32
(the IQArray layout, :229-243); the conversion kernels are in convert.hip (urhgpu_convert_dev).
"""
import ctypes as C

import numpy as np

from . import _lib

_DT = {np.dtype(np.int8): _lib.DT_I8, np.dtype(np.uint8): _lib.DT_U8, np.dtype(np.int16): _lib.DT_I16,
       np.dtype(np.uint16): _lib.DT_U16, np.dtype(np.float32): _lib.DT_F32}


def _torch_np_dtype(t):
    from .pipeline import _torch_dtype
    return _torch_dtype(t)


def _torch_dtype_for(np_dtype):
    import torch
    m = {np.dtype(np.int8): torch.int8, np.dtype(np.uint8): torch.uint8, np.dtype(np.int16): torch.int16,
         np.dtype(np.float32): torch.float32}
    if hasattr(torch, "uint16"):
        m[np.dtype(np.uint16)] = torch.uint16
    return m[np.dtype(np_dtype)]


def convert_to(data, target_dtype, ctx=None):
    """IQArray.convert_to for a device tensor (or a numpy array, which is uploaded): returns a device tensor of target_dtype."""
    import torch
    target = np.dtype(target_dtype)
    if target not in _DT:
        raise ValueError("Data type {} not supported".format(target_dtype))
    t = torch.from_numpy(np.ascontiguousarray(data)).cuda() if isinstance(data, np.ndarray) else data.contiguous()
    src = _torch_np_dtype(t)
    if src == target:
        return t
    out = torch.empty(t.shape, dtype=_torch_dtype_for(target), device=t.device)
    ctx = ctx or _lib.default_context()
    ctx.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
    _lib.check(_lib.load().urhgpu_convert_dev(ctx.handle, C.c_void_p(t.data_ptr()), _DT[src], C.c_void_p(out.data_ptr()), _DT[target],
                                              t.numel()))
    return out


def astype(data, target_dtype, ctx=None):
    """numpy's plain cast (no IQArray scaling) between float32 and an integer sample type, on the device: the raw integer values
    as float32 (Filter.apply_fir_filter, Filter.py:37-41) / the truncating write-back of IQArray.__setitem__ (IQArray.py:31-33)."""
    import torch
    target = np.dtype(target_dtype)
    t = data.contiguous()
    src = _torch_np_dtype(t)
    if src == target:
        return t
    out = torch.empty(t.shape, dtype=_torch_dtype_for(target), device=t.device)
    ctx = ctx or _lib.default_context()
    ctx.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
    _lib.check(_lib.load().urhgpu_astype_dev(ctx.handle, C.c_void_p(t.data_ptr()), _DT[src], C.c_void_p(out.data_ptr()), _DT[target],
                                             t.numel()))
    return out


def as_complex64(data, ctx=None):
    """IQArray.as_complex64 (:92-93): float32 conversion viewed as complex64 (N,)."""
    import torch
    return torch.view_as_complex(convert_to(data, np.float32, ctx).reshape(-1, 2))


def from_file(filename: str, device=None, ctx=None):
    """IQArray.from_file (:205-227): the capture is read once and uploaded; unsigned captures (.complex16u / .cu8,
    .complex32u / .cu16) become signed on the GPU, as the reference does on the host.  Returns an (N, 2) device tensor."""
    import torch
    if filename.endswith(".complex16u") or filename.endswith(".cu8"):
        raw, target = np.fromfile(filename, dtype=np.uint8), np.int8
    elif filename.endswith(".complex16s") or filename.endswith(".cs8"):
        raw, target = np.fromfile(filename, dtype=np.int8), np.int8
    elif filename.endswith(".complex32u") or filename.endswith(".cu16"):
        raw, target = np.fromfile(filename, dtype=np.uint16), np.int16
    elif filename.endswith(".complex32s") or filename.endswith(".cs16"):
        raw, target = np.fromfile(filename, dtype=np.int16), np.int16
    else:
        raw, target = np.fromfile(filename, dtype=np.float32), np.float32
    if len(raw) % 2:
        raw = raw[:-1]                                   # convert_array_to_iq drops the last half sample (:238-239)
    t = torch.from_numpy(raw.reshape(-1, 2)).to(device if device is not None else "cuda")
    return convert_to(t, target, ctx)
