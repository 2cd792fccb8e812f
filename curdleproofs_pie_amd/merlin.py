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


# ---------------------------------------------------------------------------------------------- batched, on the GPU
class TranscriptProgram:
    """A fixed sequence of transcript operations to run on MANY transcripts at once, one per GPU lane (k_merlin_batch):
    every transcript starts as `CurdleproofsTranscript(label)` and executes the same operations on its own data row.

        prog = TranscriptProgram(b"whisk_opening_proof")
        for off in range(0, 288, 48):
            prog.append(b"tracker_opening_proof", data_off=off, length=48)        # transcript.append / append_list item
        c = prog.challenge_scalar(b"tracker_opening_proof_challenge")              # get_and_append_challenge -> output slot
        outs, states = prog.run(rows)                                              # rows: one bytes object per transcript
        outs[i][c: c + 32]                                                         # the challenge of transcript i (32 B LE)
    """

    def __init__(self, label: bytes) -> None:
        self._init = ctypes.create_string_buffer(N.MERLIN_STATE_BYTES)
        N.cg1_merlin_init(self._init, bytes(label), len(label))
        self._ops: List[N.MerlinOp] = []
        self.data_bytes = 0
        self.out_bytes = 0

    def _op(self, kind: int, label: bytes, length: int, data_off: int = 0, out_off: int = 0) -> None:
        label = bytes(label)
        if len(label) > 32:
            raise ValueError("labels of the batched transcript are at most 32 bytes")
        op = N.MerlinOp(kind=kind, label_len=len(label), pad=0, len=length, data_off=data_off, out_off=out_off)
        ctypes.memmove(op.label, label, len(label))
        self._ops.append(op)

    def append(self, label: bytes, data_off: int, length: int) -> None:
        """append_message(label, row[data_off : data_off + length])"""
        self._op(0, label, length, data_off=data_off)
        self.data_bytes = max(self.data_bytes, data_off + length)

    def challenge_bytes(self, label: bytes, length: int) -> int:
        off = self.out_bytes
        self._op(1, label, length, out_off=off)
        self.out_bytes += (length + 3) & ~3
        return off

    def challenge_scalar(self, label: bytes) -> int:
        """get_and_append_challenge(label) (curdleproofs_transcript.py:15-25): returns the output offset of its 32 bytes"""
        off = self.out_bytes
        self._op(2, label, 32, out_off=off)
        self.out_bytes += 32
        return off

    def append_output(self, label: bytes, out_off: int, length: int) -> None:
        """append_message(label, <bytes the transcript produced itself at out_off>)"""
        self._op(3, label, length, out_off=out_off)

    def run(self, rows, ctx=None, want_states: bool = False):
        """rows: sequence of per-transcript data rows (bytes, each >= data_bytes long).  -> (outputs, states | None)."""
        rows = [bytes(r) for r in rows]
        n = len(rows)
        if n == 0:
            return [], ([] if want_states else None)
        stride = max(4, (max(len(r) for r in rows) + 3) & ~3)
        if any(len(r) < self.data_bytes for r in rows):
            raise ValueError("a data row is shorter than the program reads")
        ctx = ctx or N.default_context()
        ostride = max(4, self.out_bytes)
        d_data, d_out = ctx.alloc(n * stride), ctx.alloc(n * ostride)
        d_st = ctx.alloc(n * N.MERLIN_STATE_BYTES) if want_states else None
        d_data.upload(b"".join(r.ljust(stride, b"\0") for r in rows))
        ops = (N.MerlinOp * max(1, len(self._ops)))(*self._ops)
        import time

        t0 = time.perf_counter()
        ctx.check(N.cg1_merlin_batch_device(ctx.handle, self._init, ops, len(self._ops), d_data.ptr, stride, d_out.ptr, ostride,
                                            d_st.ptr if d_st else None, n))
        self.last_call_ms = (time.perf_counter() - t0) * 1e3          # the device call alone (inputs already resident)
        self.last_passes = int(N.cg1_merlin_last_passes(ctx.handle))  # Keccak passes of the slowest wave (k_merlin_batch_sync)
        out = d_out.download(n * ostride)
        outs = [out[i * ostride: (i + 1) * ostride] for i in range(n)]
        states = None
        if want_states:
            raw = d_st.download(n * N.MERLIN_STATE_BYTES)
            states = [raw[i * N.MERLIN_STATE_BYTES: (i + 1) * N.MERLIN_STATE_BYTES] for i in range(n)]
            d_st.free()
        d_data.free(); d_out.free()
        return outs, states
