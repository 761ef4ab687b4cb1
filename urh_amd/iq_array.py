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


def _raw_by_extension(name: str, read):
    """IQArray.from_file's extension table (:205-227): (flat values as stored, the IQArray's sample type).  read(dtype) -> numpy array."""
    if name.endswith(".complex16u") or name.endswith(".cu8"):
        return read(np.uint8), np.int8
    if name.endswith(".complex16s") or name.endswith(".cs8"):
        return read(np.int8), np.int8
    if name.endswith(".complex32u") or name.endswith(".cu16"):
        return read(np.uint16), np.int16
    if name.endswith(".complex32s") or name.endswith(".cs16"):
        return read(np.int16), np.int16
    return read(np.float32), np.float32


def from_file(filename: str, device=None, ctx=None):
    """IQArray.from_file (:205-227): the capture is read once and uploaded; unsigned captures (.complex16u / .cu8,
    .complex32u / .cu16) become signed on the GPU, as the reference does on the host.  A `.coco` file (Signal.__load_compressed_complex,
    Signal.py:207-213) is a tar archive whose FIRST member is such a capture, typed by the member's own name.
    Returns an (N, 2) device tensor."""
    import torch
    if filename.endswith(".coco"):
        import tarfile
        with tarfile.open(filename, "r") as tar:
            member = tar.getmembers()[0]
            blob = tar.extractfile(member).read()
        raw, target = _raw_by_extension(member.name, lambda dt: np.frombuffer(blob, dtype=dt, count=len(blob) // np.dtype(dt).itemsize).copy())
    else:
        raw, target = _raw_by_extension(filename, lambda dt: np.fromfile(filename, dtype=dt))
    if len(raw) % 2:
        raw = raw[:-1]                                   # convert_array_to_iq drops the last half sample (:238-239)
    t = torch.from_numpy(raw.reshape(-1, 2)).to(device if device is not None else "cuda")
    return convert_to(t, target, ctx)


def pcm_to_iq(frames: bytes, n_frames: int, channels: int, sample_width: int, device=None, ctx=None):
    """PCM frames -> float32 (N, 2) on the device (urhgpu_pcm_to_iq_dev: Signal.py:148-163 evaluated in float64 per sample)."""
    import torch
    dev = device if device is not None else "cuda"
    out = torch.empty((n_frames, 2), dtype=torch.float32, device=dev)
    if n_frames == 0:
        return out
    raw = torch.frombuffer(bytearray(frames), dtype=torch.uint8).to(dev)
    ctx = ctx or _lib.default_context()
    ctx.set_stream(torch.cuda.current_stream(out.device).cuda_stream)
    _lib.check(_lib.load().urhgpu_pcm_to_iq_dev(ctx.handle, C.c_void_p(raw.data_ptr()), int(n_frames), int(channels), int(sample_width),
                                                C.c_void_p(out.data_ptr())))
    out._urh_keep = raw                                  # (asynchronous: the upload lives as long as the result)
    return out


def from_wav(filename: str, device=None, ctx=None):
    """Signal.__load_wav_file (Signal.py:114-173): (iq float32 (N, 2) on the device, sample_rate, already_demodulated).  8-bit unsigned and
    16 / 24 / 32-bit signed PCM; one channel = an already demodulated capture (the samples are the real part), two = I and Q."""
    import wave
    wav = wave.open(filename, "r")
    try:
        channels, width, rate, n_frames, _, _ = wav.getparams()
        if width not in (1, 2, 3, 4):
            raise ValueError("Can't handle sample width {0}".format(width))
        frames = wav.readframes(n_frames * channels)
    finally:
        wav.close()
    if channels not in (1, 2):
        raise ValueError("Can't handle {0} channels. Only 1 and 2 are supported.".format(channels))
    if len(frames) != n_frames * channels * width:        # (the reference's assignment into its n_frames-long IQArray fails the same way)
        raise ValueError("could not broadcast input array: the file holds {0} bytes of frames, its header promises {1}".format(
            len(frames), n_frames * channels * width))
    return pcm_to_iq(frames, n_frames, channels, width, device, ctx), rate, channels == 1


def sub_file_bytes(filename: str) -> np.ndarray:
    """The uint8 samples a Flipper `.sub` file stands for (Signal.__load_sub_file, Signal.py:175-200): every `RAW_Data:` line holds run
    lengths in samples, positive = above the center (byte 255), negative = below (byte 0); values that are no integers are skipped (the
    reference logs them), a line with anything but digits, minus signs and blanks is no RAW_Data line at all.  Host parsing only."""
    import re
    runs = []
    with open(filename, "r") as fh:
        for line in fh:
            m = re.match(r"RAW_Data:\s*([-0-9 ]+)\s*$", line)
            if m:
                for value in m[1].strip().split(" "):
                    try:
                        runs.append(int(value))
                    except ValueError:
                        pass
    r = np.asarray(runs, dtype=np.int64)
    return np.repeat(np.where(r > 0, 255, 0).astype(np.uint8), np.abs(r)) if len(r) else np.zeros(0, np.uint8)


def from_sub(filename: str, device=None, ctx=None):
    """Signal.__load_sub_file (Signal.py:175-205), the Flipper Zero RAW format (OOK): the file's run lengths as bytes (sub_file_bytes),
    then (byte - 127.5) * (1 / 255) on the device like a mono 8-bit WAV.  Returns the float32 (N, 2) device tensor (real part +-0.5, an
    already demodulated capture)."""
    data = sub_file_bytes(filename)
    return pcm_to_iq(data.tobytes(), len(data), 1, 1, device, ctx)


# ---- the way out: IQArray.tofile / export_to_wav / save_compressed / export_to_sub, FileOperator.save_data -----------------------
def _converted_bytes(data, target_dtype, ctx=None) -> np.ndarray:
    """convert_to(target) on the device, then ONE copy to the host: the (N, 2) array whose bytes go into the file"""
    return convert_to(data, target_dtype, ctx).cpu().numpy()


def _target_by_extension(filename: str):
    if filename.endswith(".complex16u") or filename.endswith(".cu8"):
        return np.uint8
    if filename.endswith(".complex16s") or filename.endswith(".cs8"):
        return np.int8
    if filename.endswith(".complex32u") or filename.endswith(".cu16"):
        return np.uint16
    if filename.endswith(".complex32s") or filename.endswith(".cs16"):
        return np.int16
    return np.float32


def tofile(data, filename: str, ctx=None):
    """IQArray.tofile (:115-125): the capture converted to the sample type the extension names, raw"""
    _converted_bytes(data, _target_by_extension(filename), ctx).tofile(filename)


def export_to_wav(data, filename: str, num_channels: int, sample_rate, ctx=None):
    """IQArray.export_to_wav (:267-273): 16-bit PCM of convert_to(int16), I and Q interleaved (with num_channels = 1 the same bytes are
    2 N mono frames, as in the reference)"""
    import wave
    f = wave.open(filename, "w")
    f.setnchannels(num_channels)
    f.setsampwidth(2)
    f.setframerate(sample_rate)
    f.writeframes(_converted_bytes(data, np.int16, ctx))
    f.close()


def save_compressed(data, filename: str, ctx=None):
    """IQArray.save_compressed (:260-265): a bz2 tar archive around ONE member, the capture as float32 (the member carries no extension)"""
    import os
    import tarfile
    import tempfile
    with tarfile.open(filename, "w:bz2") as tar_write:
        tmp_name = tempfile.mkstemp()[1]
        _converted_bytes(data, np.float32, ctx).tofile(tmp_name)
        tar_write.add(tmp_name)
    os.remove(tmp_name)


def export_to_sub(data, filename: str, frequency=433920000, preset="FuriHalSubGhzPresetOok650Async", ctx=None):
    """IQArray.export_to_sub (:275-323): the uint8 conversion's first component as Flipper RAW run lengths (urhgpu_sub_encode_runs restates
    the reference's walk), 512 values per RAW_Data line"""
    u8 = _converted_bytes(data, np.uint8, ctx)
    flat = np.ascontiguousarray(u8)
    n = int(flat.shape[0])
    stride = int(flat.shape[1]) if flat.ndim > 1 else 1
    lib = _lib.load()
    cap = max(n, 1)
    runs = np.zeros(cap, dtype=np.int64)
    n_runs = C.c_int64(0)
    _lib.check(lib.urhgpu_sub_encode_runs(C.c_void_p(flat.ctypes.data), n, stride, C.c_void_p(runs.ctypes.data), cap, C.byref(n_runs)))
    arr = runs[:n_runs.value].tolist()
    with open(filename, "w") as subfile:
        subfile.write("Filetype: Flipper SubGhz RAW File\n")
        subfile.write("Version: 1\n")
        subfile.write("Frequency: {}\n".format(frequency))
        subfile.write("Preset: {}\n".format(preset))
        subfile.write("Protocol: RAW")
        for lo in range(0, len(arr), 512):
            subfile.write("\nRAW_Data: " + " ".join(map(str, arr[lo:lo + 512])))
        subfile.write("\n")


def save_data(data, filename: str, sample_rate=1e6, num_channels=2, ctx=None):
    """FileOperator.save_data (FileOperator.py:185-196) for a capture on the device (or a numpy array, which is uploaded)"""
    if filename.endswith(".wav"):
        export_to_wav(data, filename, num_channels, sample_rate, ctx)
    elif filename.endswith(".coco"):
        save_compressed(data, filename, ctx)
    elif filename.endswith(".sub"):
        export_to_sub(data, filename, ctx=ctx)
    else:
        tofile(data, filename, ctx)
