#!/usr/bin/env python3
"""Golden vectors for the capture FILE formats either side of the path (Signal.py:69-213): small WAV files (8-bit unsigned, 16 / 24 / 32-bit
signed; mono = already demodulated, stereo = I / Q), a Flipper `.sub` file and `.coco` archives are WRITTEN here (seeded) and LOADED with the
REAL reference `Signal` (oracle/ref_python.py on top of oracle/_ref); what it holds afterwards -- samples, already_demodulated, sample
rate, the automatic noise threshold, qad -- is stored in tests/golden/files/expected.npz.  Run in the build container (it needs
/root/reference):  python tests/golden/make_fileformats_golden.py"""
import io
import os
import sys
import tarfile
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "files")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))


def write_wav(path, width, channels, rate, n, rng):
    lim = {1: (0, 255), 2: (-32768, 32767), 3: (-8388608, 8388607), 4: (-2147483648, 2147483647)}[width]
    t = np.arange(n)
    cols = []
    for c in range(channels):
        env = np.repeat(rng.integers(0, 2, n // 40 + 1), 40)[:n] * 0.8 + 0.1
        x = env * np.cos(2 * np.pi * 0.03 * t + c * np.pi / 2) + 0.02 * rng.standard_normal(n)
        v = np.clip(np.round((x * 0.5 + (0.5 if width == 1 else 0.0)) * (lim[1] - lim[0]) / (1 if width == 1 else 1) * (1.0 if width == 1 else 1.0)
                             + (lim[0] if width == 1 else 0)), lim[0], lim[1]).astype(np.int64)
        v[:4] = [lim[0], lim[1], (lim[0] + lim[1]) // 2, (lim[0] + lim[1]) // 2 + 1]      # the extremes and the two values around the center
        cols.append(v)
    frames = np.stack(cols, axis=1).reshape(-1)
    if width == 1:
        raw = frames.astype(np.uint8).tobytes()
    elif width == 2:
        raw = frames.astype("<i2").tobytes()
    elif width == 4:
        raw = frames.astype("<i4").tobytes()
    else:
        raw = frames.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    w = wave.open(path, "w")
    w.setnchannels(channels); w.setsampwidth(width); w.setframerate(rate)
    w.writeframes(raw)
    w.close()


def write_sub(path, rng):
    lines = ["Filetype: Flipper SubGhz RAW File", "Version: 1", "Frequency: 433920000", "Preset: FuriHalSubGhzPresetOok650Async", "Protocol: RAW"]
    for k in range(5):
        vals, sign = [], 1
        for _ in range(int(rng.integers(20, 60))):
            vals.append(str(sign * int(rng.choice([350, 700, 352, 1050, 9000]))))
            sign = -sign
        if k == 2:
            vals.insert(3, "x12")           # no RAW_Data line any more for the reference's pattern: the whole line is dropped
        if k == 3:
            vals.insert(5, "")              # a double space: int("") fails, the value is skipped
            vals.insert(9, "0")             # zero samples
        lines.append("RAW_Data: " + " ".join(vals))
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def write_coco(path, member_name, arr):
    blob = arr.tobytes()
    with tarfile.open(path, "w:bz2") as tar:
        info = tarfile.TarInfo(member_name)
        info.size = len(blob)
        tar.addfile(info, io.BytesIO(blob))


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260924)
    names = []
    for width, channels, rate in ((1, 1, 8000), (2, 1, 44100), (3, 1, 48000), (4, 1, 1000000), (1, 2, 2000000), (2, 2, 1000000), (3, 2, 250000), (4, 2, 96000)):
        name = f"pcm{8 * width}_{'mono' if channels == 1 else 'stereo'}.wav"
        write_wav(os.path.join(OUT, name), width, channels, rate, 3001, rng)
        names.append(name)
    write_sub(os.path.join(OUT, "flipper.sub"), rng)
    names.append("flipper.sub")
    n = 6000
    ph = 2 * np.pi * np.cumsum(np.repeat(np.where(rng.integers(0, 2, n // 50 + 1) == 1, 0.02, -0.02), 50)[:n])
    iq = (np.stack([np.cos(ph), np.sin(ph)], 1) + 0.03 * rng.standard_normal((n, 2))).astype(np.float32)
    write_coco(os.path.join(OUT, "float.coco"), "tmp/tmpfile", iq)
    write_coco(os.path.join(OUT, "signed8.coco"), "capture.complex16s", np.clip(np.round(iq * 100), -128, 127).astype(np.int8))
    write_coco(os.path.join(OUT, "unsigned16.coco"), "capture.cu16", np.clip(np.round(iq * 20000 + 32768), 0, 65535).astype(np.uint16))
    names += ["float.coco", "signed8.coco", "unsigned16.coco"]

    import ref_python
    ref_python.setup()
    from urh.signalprocessing.Signal import Signal
    out = {}
    for name in names:
        s = Signal(os.path.join(OUT, name), name)
        out[name + "/iq"] = np.ascontiguousarray(s.iq_array.data)
        out[name + "/already_demodulated"] = np.array(bool(s.already_demodulated))
        out[name + "/sample_rate"] = np.array(float(s.sample_rate))
        out[name + "/noise_threshold"] = np.array(float(s.noise_threshold))
        s.modulation_type = "FSK"
        out[name + "/qad"] = np.ascontiguousarray(s.qad, dtype=np.float32)
        print(name, s.iq_array.data.dtype, s.iq_array.data.shape, bool(s.already_demodulated), s.sample_rate, s.noise_threshold)
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **out)


if __name__ == "__main__":
    main()
