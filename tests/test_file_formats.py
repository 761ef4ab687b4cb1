"""Capture FILE formats either side of the path (Signal.py:69-213): WAV (8-bit unsigned, 16 / 24 / 32-bit signed PCM; mono = already
demodulated, stereo = I / Q), Flipper `.sub` and `.coco` archives through `urh_amd.signal.Signal.from_file`, against what the REAL
reference `Signal` holds after loading the same files (tests/golden/files/expected.npz, written by make_fileformats_golden.py): the
samples bit for bit, already_demodulated, the sample rate, the automatic noise threshold (Signal.py:97-107) and qad."""
import os
import wave

import numpy as np
import pytest

from conftest import ROOT

FILES = os.path.join(ROOT, "tests", "golden", "files")
NAMES = ["pcm8_mono.wav", "pcm16_mono.wav", "pcm24_mono.wav", "pcm32_mono.wav", "pcm8_stereo.wav", "pcm16_stereo.wav", "pcm24_stereo.wav",
         "pcm32_stereo.wav", "flipper.sub", "float.coco", "signed8.coco", "unsigned16.coco"]


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_signal_from_file_equals_the_reference(name):
    from urh_amd.signal import Signal
    exp = np.load(os.path.join(FILES, "expected.npz"))
    s = Signal.from_file(os.path.join(FILES, name), default_noise_threshold="automatic")
    iq = s.iq.cpu().numpy() if hasattr(s.iq, "cpu") else np.asarray(s.iq)
    want = exp[name + "/iq"]
    assert iq.dtype == want.dtype and iq.shape == want.shape and np.array_equal(_bits(iq), _bits(want)), (name, iq.dtype, want.dtype, iq.shape, want.shape)
    assert s.already_demodulated == bool(exp[name + "/already_demodulated"])
    assert float(s.sample_rate) == float(exp[name + "/sample_rate"])
    assert float(s.noise_threshold) == float(exp[name + "/noise_threshold"]), (name, s.noise_threshold, float(exp[name + "/noise_threshold"]))
    s.modulation_type = "FSK"
    q = np.asarray(s.qad_host(), dtype=np.float32)
    assert np.array_equal(_bits(q), _bits(exp[name + "/qad"])), name
    if s.already_demodulated:
        assert s.demod_passes == 0                        # the real part IS the demodulated signal (Signal.py:424-427)


@pytest.mark.gpu
def test_percent_noise_threshold_setting():
    """default_noise_threshold as a number: that many percent of max_magnitude (Signal.py:104-107)"""
    from urh_amd.signal import Signal
    s = Signal.from_file(os.path.join(FILES, "signed8.coco"), default_noise_threshold=5)
    assert s.noise_threshold == 5 / 100 * (2 * 128 ** 2) ** 0.5
    s = Signal.from_file(os.path.join(FILES, "float.coco"))
    assert s.noise_threshold == 0 and s.filename.endswith("float.coco") and not s.wav_mode and not s.flipper_raw_mode


def test_wav_with_three_channels_is_refused(tmp_path):
    """the reference's ValueError (Signal.py:164-169); raised before anything touches the device"""
    from urh_amd import iq_array
    f = str(tmp_path / "three.wav")
    w = wave.open(f, "w")
    w.setnchannels(3); w.setsampwidth(2); w.setframerate(8000)
    w.writeframes(np.zeros(30, "<i2").tobytes())
    w.close()
    with pytest.raises(ValueError, match="Can't handle 3 channels"):
        iq_array.from_wav(f)


# ---- the way out: FileOperator.save_data (FileOperator.py:185-196) ---------------------------------------------------------------
EXPORT_EXTS = [".complex", ".complex16u", ".cu8", ".complex16s", ".cs8", ".complex32u", ".cu16", ".complex32s", ".cs16", ".wav", ".sub", ".coco"]


@pytest.mark.gpu
@pytest.mark.parametrize("ext", EXPORT_EXTS)
def test_save_data_writes_the_reference_bytes(tmp_path, ext):
    """every sample type saved under every extension: the file's bytes (for `.coco`: the archive member's) are those the reference's
    IQArray.tofile / export_to_wav / save_compressed / export_to_sub wrote for the same capture (make_export_golden.py)"""
    import tarfile
    from urh_amd import iq_array
    exp = np.load(os.path.join(FILES, "export_expected.npz"))
    names = [k[3:] for k in exp.files if k.startswith("in/")]
    assert len(names) == 15
    for name in names:
        arr = exp["in/" + name]
        for ch in ((1, 2) if ext == ".wav" else (2,)):
            f = str(tmp_path / (name + ext))
            iq_array.save_data(arr, f, sample_rate=250000.0, num_channels=ch)
            if ext == ".coco":
                with tarfile.open(f, "r") as tar:
                    assert len(tar.getmembers()) == 1
                    blob = tar.extractfile(tar.getmembers()[0]).read()
            else:
                blob = open(f, "rb").read()
            want = exp[f"out/{name}{ext}" + (f"/ch{ch}" if ext == ".wav" else "")].tobytes()
            assert blob == want, (name, ext, ch, len(blob), len(want))
            os.remove(f)


@pytest.mark.gpu
def test_saved_files_load_back(tmp_path):
    """save_data -> Signal.from_file round trips: raw float32 and `.coco` exactly, `.sub` as the +-0.5 square wave of its runs"""
    from urh_amd import iq_array
    from urh_amd.signal import Signal
    exp = np.load(os.path.join(FILES, "export_expected.npz"))
    arr = exp["in/tone_f32"]
    for ext in (".complex", ".coco"):
        f = str(tmp_path / ("tone" + ext))
        iq_array.save_data(arr, f)
        assert np.array_equal(Signal.from_file(f).iq.cpu().numpy().view(np.uint32), arr.view(np.uint32))
    sig = Signal(exp["in/tone_i16"])
    sig.changed = True
    sig.filename = str(tmp_path / "renamed.complex32s")
    sig.save()                                             # Signal.save -> save_as -> FileOperator.save_signal (Signal.py:462-472)
    assert not sig.changed and sig.name == "renamed" and open(sig.filename, "rb").read() == exp["out/tone_i16.complex32s"].tobytes()
    f = str(tmp_path / "clean.sub")
    iq_array.save_data(exp["in/clean_f32"], f)
    s = Signal.from_file(f)
    back = s.iq.cpu().numpy()
    assert s.already_demodulated and set(np.unique(back[:, 0]).tolist()) <= {-0.5, 0.5} and not back[:, 1].any()
    assert abs(len(back) - len(arr)) <= 2


def _reference_sub_walk(values):
    """IQArray.export_to_sub's loop (IQArray.py:279-304), restated for the checker"""
    arr, counter, last = [], 0, None
    for v in values:
        if last is None:
            last = v
        if v == last:
            counter += 1
        elif counter > 1:
            arr.append(counter if last > 127 else -counter)
            counter = 1
            last = v
    arr.append(counter if last > 127 else -counter)
    return arr


def test_sub_run_encoder_equals_the_reference_walk():
    """urhgpu_sub_encode_runs (host arithmetic) on random byte streams: long runs, single-sample runs (which never end a run), strides"""
    import ctypes as C
    from urh_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for it in range(300):
        n = int(rng.integers(1, 400))
        kinds = int(rng.choice([2, 3, 5]))
        vals = rng.choice(np.array([0, 255, 127, 128, 200], np.uint8)[:kinds], size=n)
        v = np.repeat(vals, rng.integers(1, 4, size=n))[:n].astype(np.uint8)
        stride = int(rng.choice([1, 2]))
        buf = np.zeros((len(v), stride), np.uint8)
        buf[:, 0] = v
        want = _reference_sub_walk(v.tolist())
        out = np.zeros(len(v) + 1, np.int64)
        k = C.c_int64(0)
        assert lib.urhgpu_sub_encode_runs(buf.ctypes.data_as(C.c_void_p), len(v), stride, out.ctypes.data_as(C.c_void_p), len(out), C.byref(k)) == 0
        assert out[:k.value].tolist() == want, (it, v.tolist(), want, out[:k.value].tolist())
    k = C.c_int64(0)
    assert lib.urhgpu_sub_encode_runs(buf.ctypes.data_as(C.c_void_p), 0, 1, None, 0, C.byref(k)) == _lib.ERR_ARG


def test_sub_file_parser_equals_the_reference_samples():
    """the host half of the `.sub` loader: the bytes the run lengths stand for, against the samples the real Signal held after loading the
    same file (+0.5 where the byte is 255, -0.5 where it is 0) -- incl. the line the reference's pattern rejects, the empty value of a
    double blank and the zero-length run (make_fileformats_golden.py)"""
    from urh_amd import iq_array
    exp = np.load(os.path.join(FILES, "expected.npz"))
    want = exp["flipper.sub/iq"]
    got = iq_array.sub_file_bytes(os.path.join(FILES, "flipper.sub"))
    assert got.dtype == np.uint8 and len(got) == len(want) and set(np.unique(got).tolist()) <= {0, 255}
    assert np.array_equal(np.where(got == 255, np.float32(0.5), np.float32(-0.5)), want[:, 0]) and not want[:, 1].any()
