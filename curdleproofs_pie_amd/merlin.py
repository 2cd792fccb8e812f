"""Native Merlin transcript with the reference's Python API.

    from curdleproofs_pie_amd.merlin import MerlinTranscript, Strobe128, CurdleproofsTranscript

Mirrors /root/reference/merlin_transcripts/merlin_transcripts/{merlin_transcript.py, strobe.py} and
/root/reference/curdleproofs/curdleproofs/curdleproofs_transcript.py: same class and method names, same
arguments, same outputs; the sponge runs in libcurdle_g1.so instead of pure Python."""
from __future__ import annotations

import ctypes
from typing import List

from . import _native as N
from .py_arkworks_bls12381 import Scalar


class Strobe128:  # strobe.py:16-107 (the subset Merlin uses)
    def __init__(self, _state=None) -> None:
        self._st = _state if _state is not None else ctypes.create_string_buffer(N.MERLIN_STATE_BYTES)

    @classmethod
    def new(cls, protocol_label: bytes) -> "Strobe128":
        s = cls()
        N.cg1_strobe_new(s._st, bytes(protocol_label), len(protocol_label))
        return s

    @staticmethod
    def _chk(rc: int) -> None:
        assert rc == 0  # the reference asserts `cur_flags == flags` for more=True (strobe.py:91)

    def meta_ad(self, data: bytes, more: bool) -> None:
        self._chk(N.cg1_strobe_meta_ad(self._st, bytes(data), len(data), int(more)))

    def ad(self, data: bytes, more: bool) -> None:
        self._chk(N.cg1_strobe_ad(self._st, bytes(data), len(data), int(more)))

    def prf(self, data_len: int, more: bool) -> bytearray:
        out = ctypes.create_string_buffer(max(data_len, 1))
        self._chk(N.cg1_strobe_prf(self._st, out, data_len, int(more)))
        return bytearray(out.raw[:data_len])

    def key(self, data: bytes, more: bool) -> None:
        self._chk(N.cg1_strobe_key(self._st, bytes(data), len(data), int(more)))


class MerlinTranscript:  # merlin_transcript.py:6-24
    def __init__(self, label: bytes) -> None:
        self.strobe = Strobe128()
        N.cg1_merlin_init(self.strobe._st, bytes(label), len(label))

    def append_message(self, label: bytes, message: bytes) -> None:
        message = bytes(message)
        N.cg1_merlin_append(self.strobe._st, bytes(label), len(label), message, len(message))

    def append_u64(self, label: bytes, x: int) -> None:
        self.append_message(label, x.to_bytes(8, "little"))

    def challenge_bytes(self, label: bytes, length: int) -> bytes:
        out = ctypes.create_string_buffer(max(length, 1))
        N.cg1_merlin_challenge(self.strobe._st, bytes(label), len(label), out, length)
        return out.raw[:length]


class CurdleproofsTranscript(MerlinTranscript):  # curdleproofs_transcript.py:7-28
    def append(self, label: bytes, item: bytes) -> None:
        self.append_message(label, item)

    def append_list(self, label: bytes, items: List[bytes]) -> None:
        items = [bytes(i) for i in items]
        if items and all(len(i) == len(items[0]) for i in items):
            N.cg1_merlin_append_list(self.strobe._st, bytes(label), len(label), b"".join(items), len(items[0]), len(items))
        else:
            for item in items:
                self.append_message(label, item)

    def get_and_append_challenge(self, label: bytes) -> Scalar:
        out = ctypes.create_string_buffer(32)
        N.cg1_merlin_challenge_scalar(self.strobe._st, bytes(label), len(label), out)
        return Scalar.from_le_bytes(out.raw)

    def get_and_append_challenges(self, label: bytes, n: int) -> List[Scalar]:
        return [self.get_and_append_challenge(label) for _ in range(0, n)]
