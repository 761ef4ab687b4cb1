"""Device-resident counterpart of ProtocolAnalyzer.get_protocol_from_signal
(/root/reference/src/urh/signalprocessing/ProtocolAnalyzer.py:227-321): the O(N) work (demodulation, pulse table, bit
expansion) is one urhgpu_iq_to_bits_dev pass; what remains per message -- padding ASK messages to a multiple of
message_length_divisor (:289-321), the RSSI over one symbol at the middle bit (:267-269), the timestamp (:270-272) -- is
host arithmetic on a few values, with the RSSI samples fetched from the GPU in one gather.
"""
import array
from dataclasses import dataclass

import numpy as np


@dataclass
class MessageData:
    """The fields ProtocolAnalyzer hands to urh's Message constructor (:273-283)."""
    plain_bits: array.array          # array('B')
    pause: int
    bit_sample_pos: array.array      # array('L')
    rssi: float
    timestamp: float
    samples_per_symbol: int
    bits_per_symbol: int

    @property
    def plain_bits_str(self):
        return "".join(map(str, self.plain_bits))


def _min_max_for_dtype(dtype):
    """IQArray.min_max_for_dtype (src/urh/signalprocessing/IQArray.py:246-250)"""
    dtype = np.dtype(dtype)
    if dtype.kind == "f" or dtype.kind == "c":
        return -1, 1
    return np.iinfo(dtype).min, np.iinfo(dtype).max


def ensure_message_length_multiple(bit_data, samples_per_symbol, pauses, bit_sample_pos, divisor):
    """ProtocolAnalyzer.__ensure_message_length_multiple (:289-321): ASK messages borrow zero bits from the pause
    that follows them so that their length becomes a multiple of `divisor`."""
    for i in range(len(bit_data)):
        missing_bits = (divisor - (len(bit_data[i]) % divisor)) % divisor
        if missing_bits > 0 and pauses[i] >= samples_per_symbol * missing_bits:
            bit_data[i].extend([0] * missing_bits)
            pauses[i] = pauses[i] - missing_bits * samples_per_symbol
            try:
                bit_sample_pos[i][-1] = bit_sample_pos[i][-2] + samples_per_symbol
            except IndexError:
                continue
            bit_sample_pos[i].extend([bit_sample_pos[i][-1] + (k + 1) * samples_per_symbol for k in range(missing_bits - 1)])
            bit_sample_pos[i].append(bit_sample_pos[i][-1] + pauses[i])


def get_protocol_from_signal_dev(pipe, iq, p, message_length_divisor=1, sample_rate=1e6, timestamp=0.0):
    """iq: capture on the GPU ((N, 2) tensor of a supported dtype or complex64 (N,)); p: pipeline.DemodParams.
    Returns the list of MessageData the reference would build its Message objects from."""
    torch = pipe.torch
    if iq.dtype == torch.complex64:
        iq = torch.view_as_real(iq)
    res = pipe.iq_to_bits_checked(iq, p, want_qad=True)
    bit_data, pauses, bit_sample_pos = res.messages()
    return messages_from_bits(pipe, iq, p, bit_data, pauses, bit_sample_pos, message_length_divisor, sample_rate, timestamp)


def messages_from_bits(pipe, iq, p, bit_data, pauses, bit_sample_pos, message_length_divisor=1, sample_rate=1e6, timestamp=0.0):
    """The per-message part of get_protocol_from_signal (:256-283) for bits that are already on the host."""
    torch = pipe.torch
    sps = int(p.samples_per_symbol)
    if message_length_divisor > 1 and p.modulation_type == "ASK":
        ensure_message_length_multiple(bit_data, sps, pauses, bit_sample_pos, int(message_length_divisor))
    n = int(iq.shape[0])
    # RSSI: mean of magnitudes_normalized over [middle_bit_pos, middle_bit_pos + samples_per_symbol) -- one gather
    starts = [int(bit_sample_pos[i][int(len(bits) / 2)]) for i, bits in enumerate(bit_data)]
    messages = []
    if starts:
        idx = torch.tensor([s + k for s in starts for k in range(sps) if s + k < n], dtype=torch.int64, device=iq.device)
        got = iq[idx].cpu().numpy()
        lo, hi = _min_max_for_dtype(got.dtype)
        norm = np.sqrt(hi ** 2.0 + lo ** 2.0)
        off = 0
        for i, (bits, pause) in enumerate(zip(bit_data, pauses)):
            cnt = max(0, min(starts[i] + sps, n) - starts[i])
            sl = got[off:off + cnt]
            off += cnt
            if sl.dtype == np.float32:                                   # util.get_magnitudes: fp32 sqrtf, stored as float64
                mags = np.sqrt(sl[:, 0] * sl[:, 0] + sl[:, 1] * sl[:, 1]).astype(np.float64)
            else:                                                        # integer dtypes: C int arithmetic, double sqrt
                a = sl.astype(np.int64)
                s32 = ((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
                with np.errstate(invalid="ignore"):
                    mags = np.sqrt(s32.astype(np.float64))
            with np.errstate(invalid="ignore", divide="ignore"):
                rssi = float(np.mean(mags / norm)) if cnt else float("nan")
            messages.append(MessageData(bits, int(pause), bit_sample_pos[i], rssi,
                                        timestamp + bit_sample_pos[i][0] / sample_rate, sps, int(p.bits_per_symbol)))
    return messages
