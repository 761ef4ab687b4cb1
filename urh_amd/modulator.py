"""`Modulator`: the caller of modulate_c on the other side of the path (Generator tab, urh_cli TX) -- the reference's
/root/reference/src/urh/signalprocessing/Modulator.py without its Qt scene, XML persistence and display strings: parameters per
modulation type, `modulate()` (:215-255) on the GPU through urh_amd.signal_functions.modulate_c (modulate.hip), the default
parameters of a modulation order (:257-283), the carrier guess from a Signal (:307-318).
"""
import array
import math

import numpy as np

from . import signal_functions


class Modulator(object):
    FORCE_DTYPE = None
    DEFAULT_DTYPE = np.float32          # the reference reads it from its settings ("modulation_dtype", :64-75)

    MODULATION_TYPES = ["ASK", "FSK", "PSK", "GFSK", "OQPSK"]

    def __init__(self, name: str):
        self.carrier_freq_hz = 40 * 10 ** 3
        self.carrier_amplitude = 1
        self.carrier_phase_deg = 0
        self.data = [True, False, True, False]
        self.samples_per_symbol = 100
        self.default_sample_rate = 10 ** 6
        self._sample_rate = None
        self._modulation_type = "ASK"
        self._bits_per_symbol = 1
        self.name = name
        self.gauss_bt = 0.5                  # bt product of the Gaussian filter (GFSK)
        self.gauss_filter_width = 1
        self.parameters = array.array("f", [0, 100])      # frequencies, amplitudes (0 .. 100 %) or phases (0 .. 360)

    def __eq__(self, other):
        return (self.carrier_freq_hz == other.carrier_freq_hz and self.carrier_amplitude == other.carrier_amplitude
                and self.carrier_phase_deg == other.carrier_phase_deg and self.name == other.name
                and self.modulation_type == other.modulation_type and self.samples_per_symbol == other.samples_per_symbol
                and self.bits_per_symbol == other.bits_per_symbol and self.sample_rate == other.sample_rate
                and self.parameters == other.parameters)

    @staticmethod
    def get_dtype():
        return Modulator.FORCE_DTYPE if Modulator.FORCE_DTYPE is not None else Modulator.DEFAULT_DTYPE

    @property
    def modulation_type(self) -> str:
        return self._modulation_type

    @modulation_type.setter
    def modulation_type(self, value):
        try:
            self._modulation_type = self.MODULATION_TYPES[int(value)]       # (legacy: the type saved as an index, :81-87)
        except (ValueError, IndexError):
            self._modulation_type = value

    @property
    def is_binary_modulation(self):
        return self.bits_per_symbol == 1

    @property
    def is_amplitude_based(self):
        return "ASK" in self.modulation_type

    @property
    def is_frequency_based(self):
        return "FSK" in self.modulation_type

    @property
    def is_phase_based(self):
        return "PSK" in self.modulation_type

    @property
    def bits_per_symbol(self):
        return self._bits_per_symbol

    @bits_per_symbol.setter
    def bits_per_symbol(self, value):
        value = int(value)
        if value != self.bits_per_symbol:
            self._bits_per_symbol = value
            self.parameters = array.array("f", [0] * self.modulation_order)

    @property
    def modulation_order(self):
        return 2 ** self.bits_per_symbol

    @property
    def sample_rate(self):
        return self._sample_rate if self._sample_rate is not None else self.default_sample_rate

    @sample_rate.setter
    def sample_rate(self, value):
        self._sample_rate = value

    def modulate(self, data=None, pause=0, start=0, dtype=None) -> np.ndarray:
        """Modulator.modulate (:215-255): bits (a str of 0 / 1, a list, an array) -> (N, 2) IQ samples of `dtype`; amplitudes are percent
        of the sample type's maximum, phases degrees.  Returns the samples as a numpy array (the reference wraps them in an IQArray)."""
        assert pause >= 0
        if data is None:
            data = self.data
        else:
            self.data = data
        if isinstance(data, str):
            data = array.array("B", map(int, data))
        elif isinstance(data, list):
            data = array.array("B", data)
        if len(data) == 0:
            return np.zeros((0, 2), dtype=np.float32)
        dtype = dtype or self.get_dtype()
        dt = np.dtype(dtype)
        a = self.carrier_amplitude * (1 if dt.kind == "f" else np.iinfo(dt).max)
        parameters = self.parameters
        if self.modulation_type == "ASK":
            parameters = array.array("f", [a * p / 100 for p in parameters])
        elif self.modulation_type == "PSK":
            parameters = array.array("f", [p * (math.pi / 180) for p in parameters])
        return signal_functions.modulate_c(data, self.samples_per_symbol, self.modulation_type, parameters, self.bits_per_symbol, a,
                                           self.carrier_freq_hz, self.carrier_phase_deg * (np.pi / 180), self.sample_rate, pause, start,
                                           dtype, self.gauss_bt, self.gauss_filter_width)

    def get_default_parameters(self) -> array.array:
        """Modulator.get_default_parameters (:257-277)"""
        if self.is_amplitude_based:
            parameters = np.linspace(0, 100, self.modulation_order, dtype=np.float32)
        elif self.is_frequency_based:
            parameters = [(i + 1) * self.carrier_freq_hz / self.modulation_order for i in range(self.modulation_order)]
        elif self.is_phase_based:
            step = 360 / self.modulation_order
            parameters = np.arange(step / 2, 360, step) - 180
            if self.modulation_type == "OQPSK":
                parameters = parameters[[i ^ (i >> 1) for i in range(self.modulation_order)]]      # Gray code order (:279-283)
        else:
            return None
        return array.array("f", parameters)

    def estimate_carrier_frequency(self, signal, start: int, num_samples: int):
        """Modulator.estimate_carrier_frequency (:307-318) for the first message's sample range (the reference asks its ProtocolAnalyzer
        for it): at most 10^6 samples, through Signal.estimate_frequency"""
        if num_samples > 1e6:
            num_samples = int(1e6)
        return signal.estimate_frequency(start, start + num_samples, self.sample_rate)
