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
