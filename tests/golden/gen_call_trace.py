#!/usr/bin/env python3
"""Record the SEQUENCE OF BACKEND CALLS the reference makes for one Whisk shuffle proof at ell = 124 (N = 128): one
GenerateWhiskShuffleProof (whisk_interface.py:111-144 -> CurdleProofsProof.new, curdleproofs.py:50-160) and one
IsValidWhiskShuffleProof (whisk_interface.py:72-109 -> CurdleProofsProof.verify, curdleproofs.py:162-248), unmodified, in the
build container, over a TRACING stand-in for the Rust wheel.  What crosses the boundary SURVEY.md 8(b) describes is logged:

  G1Point operators (generator, identity, + - neg, * Scalar, ==, to_compressed_bytes, from_compressed_bytes[_unchecked]),
  compute_MSM(bases, scalars) and MSMAccumulator.accumulate_check / verify  (msm_accumulator.py:6-12, :37-68; ONE record each --
  their inner loops run untraced),

with operands as value numbers (every G1 result gets the next id), scalars by value, and the outputs a caller can observe
(compressed bytes, equality results, the verifier's verdict, the random factor each accumulate_check drew) recorded next to the
call.  Values are computed by the pure-Python oracle backend (tests/golden/_backend.py), so no expected byte comes out of product
arithmetic.  Scalar arithmetic and the Merlin transcript are the callers' own Python and are not part of the trace.

Data only:
  call_trace_ell124.json   header + three op lists: "setup" (CRS + trackers from bytes), "prove", "verify"
  call_trace_ell124.bin    the scalars (32 B little-endian each) and 48-byte encodings the ops refer to by index

tools/replay_call_trace.py replays the lists through the product's Python face (curdleproofs_pie_amd) on the GPU box: every
recorded output must come back bit for bit, and the wall time of the replay is the cost of the reference's own, unchanged control
flow on this backend.

    python tests/golden/gen_call_trace.py [--backend oracle|product] [--out DIR]
"""
import hashlib
import json
import os
import random
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _backend  # noqa: E402

ELL, N_BLINDERS = 124, 4
CRS_SEED, PROOF_SEED = 9000, 9100          # the CRS and proof 0 of tests/golden/shuffle_batch_ell124.bin


class Trace:
    def __init__(self):
        self.ops = []
        self.blob = bytearray()
        self.next_id = 0
        self.on = True
        self.rhos = []

    def new_id(self):
        i = self.next_id
        self.next_id += 1
        return i

    def scalar(self, s):
        off = len(self.blob) // 16
        self.blob += int(s).to_bytes(32, "little")
        return off

    def enc(self, b):
        b = bytes(b)
        assert len(b) == 48
        off = len(self.blob) // 16
        self.blob += b
        return off


T = Trace()


def make_tracing_backend(inner):
    """A module with the wheel's two names: Scalar is the inner backend's own class, G1Point a logging wrapper around its points."""
    IP = inner.G1Point

    class G1Point:
        __slots__ = ("_p", "_id")
        __hash__ = None

        def __init__(self):
            self._p = IP()
            self._id = T.new_id() if T.on else -1
            if T.on:
                T.ops.append(["gen", self._id])

        @staticmethod
        def _wrap(p, op, *args):
            o = object.__new__(G1Point)
            o._p = p
            o._id = -1
            if T.on:
                o._id = T.new_id()
                T.ops.append([op, o._id] + list(args))
            return o

        @staticmethod
        def _id_of(x):
            if x._id < 0:
                raise RuntimeError("a point made while tracing was suspended leaked into a traced call")
            return x._id

        @staticmethod
        def identity():
            return G1Point._wrap(IP.identity(), "id")

        def __add__(self, o):
            return G1Point._wrap(self._p + o._p, "add", *( [G1Point._id_of(self), G1Point._id_of(o)] if T.on else []))

        __radd__ = __add__

        def __sub__(self, o):
            return G1Point._wrap(self._p - o._p, "sub", *( [G1Point._id_of(self), G1Point._id_of(o)] if T.on else []))

        def __neg__(self):
            return G1Point._wrap(-self._p, "neg", *( [G1Point._id_of(self)] if T.on else []))

        def __mul__(self, s):
            return G1Point._wrap(self._p * s, "mul", *( [G1Point._id_of(self), T.scalar(s)] if T.on else []))

        __rmul__ = __mul__

        def __eq__(self, o):
            r = isinstance(o, G1Point) and self._p == o._p
            if T.on and isinstance(o, G1Point):
                T.ops.append(["eq", G1Point._id_of(self), G1Point._id_of(o), bool(r)])
            return r

        def __ne__(self, o):
            return not self.__eq__(o)

        def to_compressed_bytes(self):
            b = bytes(self._p.to_compressed_bytes())
            if T.on:
                T.ops.append(["cmp", G1Point._id_of(self), T.enc(b)])
            return b

        def __str__(self):
            return bytes(self._p.to_compressed_bytes()).hex()

        @staticmethod
        def from_compressed_bytes(data):
            p = IP.from_compressed_bytes(bytes(data))
            return G1Point._wrap(p, "dec", *( [T.enc(data), 1] if T.on else []))

        @staticmethod
        def from_compressed_bytes_unchecked(data):
            p = IP.from_compressed_bytes_unchecked(bytes(data))
            return G1Point._wrap(p, "dec", *( [T.enc(data), 0] if T.on else []))

    m = types.ModuleType("py_arkworks_bls12381")
    m.G1Point = G1Point
    m.Scalar = inner.Scalar
    m.__inner__ = inner.__name__
    return m


class suspended:
    def __enter__(self):
        self.was = T.on
        T.on = False

    def __exit__(self, *a):
        T.on = self.was


def main():
    if _backend.BACKEND == "oracle":
        import oracle.py_arkworks_shim as inner
    else:
        import curdleproofs_pie_amd.py_arkworks_bls12381 as inner
    tb = make_tracing_backend(inner)
    sys.modules["py_arkworks_bls12381"] = tb
    T.on = False                                      # module-level singletons (util.G1, util.Z1) are made at import
    import curdleproofs.msm_accumulator as MA
    import curdleproofs.util as U

    G1Point = tb.G1Point
    # util.G1 / util.Z1 are module constants every caller shares: give them value numbers in the setup list
    T.on = True
    U.G1 = G1Point()
    U.Z1 = G1Point.identity()
    MA.Z1 = U.Z1
    T.on = False

    orig_acc = MA.MSMAccumulator
    orig_random_scalar = MA.random_scalar

    def traced_compute_MSM(bases, scalars):          # msm_accumulator.py:6-12, as ONE record
        pairs = list(zip(bases, scalars))
        with suspended():
            cur = inner.G1Point.identity()
            for b, s in pairs:
                cur = cur + b._p * s
        return G1Point._wrap(cur, "msm", *( [[G1Point._id_of(b) for b, _ in pairs], [T.scalar(s) for _, s in pairs]] if T.on else []))

    class TracedAccumulator(orig_acc):                # msm_accumulator.py:32-68: one record per call, the reference's own body inside
        def __init__(self):
            with suspended():
                super().__init__()
            self._tid = None
            if T.on:
                self._tid = sum(1 for o in T.ops if o[0] == "acc_new")
                T.ops.append(["acc_new", self._tid])

        def accumulate_check(self, C, bases, scalars):
            bases, scalars = list(bases), list(scalars)
            rec = None
            if T.on:
                rec = ["acc_check", self._tid, G1Point._id_of(C), [G1Point._id_of(b) for b in bases], [T.scalar(s) for s in scalars], None]
                T.ops.append(rec)
            drawn = []

            def rs():
                r = orig_random_scalar()
                drawn.append(r)
                return r

            MA.random_scalar = rs
            try:
                with suspended():
                    super().accumulate_check(C, bases, scalars)
            finally:
                MA.random_scalar = orig_random_scalar
            assert len(drawn) == 1                    # msm_accumulator.py:43: exactly one draw per call
            if rec is not None:
                rec[5] = T.scalar(drawn[0])

        def verify(self):
            ok = True
            try:
                with suspended():
                    super().verify()
            except AssertionError:
                ok = False
            if T.on:
                T.ops.append(["acc_verify", self._tid, ok])
            if not ok:
                raise AssertionError()

    MA.compute_MSM = traced_compute_MSM
    MA.MSMAccumulator = TracedAccumulator
    orig_compute = None

    # now the callers (they bind compute_MSM / MSMAccumulator / G1 / Z1 at import)
    from curdleproofs.crs import CurdleproofsCrs
    from curdleproofs.util import BLSPubkey, BufReader, point_projective_to_bytes, random_scalar
    from curdleproofs.whisk_interface import GenerateWhiskShuffleProof, IsValidWhiskShuffleProof, WhiskTracker

    # every caller module binds these names at ITS import (`from curdleproofs.util import G1, Z1`, `from curdleproofs.msm_accumulator
    # import MSMAccumulator, compute_MSM`): point all of them at the traced objects, whatever the import order was
    for name, mod in list(sys.modules.items()):
        if not name.startswith("curdleproofs") or mod is None:
            continue
        for attr, val in (("G1", U.G1), ("Z1", U.Z1), ("compute_MSM", traced_compute_MSM), ("MSMAccumulator", TracedAccumulator)):
            if hasattr(mod, attr) and getattr(mod, attr) is not val:
                setattr(mod, attr, val)

    # ---- setup (untraced arithmetic; only what the replay needs as inputs is logged: the CRS and the trackers arrive as bytes)
    random.seed(CRS_SEED)
    with suspended():
        crs_bytes = bytes(CurdleproofsCrs.new(ELL, N_BLINDERS).to_bytes())
        random.seed(PROOF_SEED)
        pre = []
        for _ in range(ELL):
            k, r = random_scalar(), random_scalar()
            r_G = U.G1 * r
            pre.append(WhiskTracker(BLSPubkey(point_projective_to_bytes(r_G)), BLSPubkey(point_projective_to_bytes(r_G * k))))
    T.on = True
    crs = CurdleproofsCrs.from_bytes(BufReader(crs_bytes), ELL, N_BLINDERS)          # 133 "dec" records
    n_setup = len(T.ops)

    # ---- prove
    post, proof = GenerateWhiskShuffleProof(crs, pre)
    n_prove = len(T.ops)

    # ---- verify
    ok = IsValidWhiskShuffleProof(crs, pre, post, proof)
    assert ok
    T.on = False

    ops = T.ops
    out_dir = _backend.OUT or os.path.dirname(os.path.abspath(__file__))
    blob = bytes(T.blob)
    with open(os.path.join(out_dir, "call_trace_ell124.bin"), "wb") as f:
        f.write(blob)
    counts = {}
    for phase, lo, hi in (("setup", 0, n_setup), ("prove", n_setup, n_prove), ("verify", n_prove, len(ops))):
        c = {}
        for o in ops[lo:hi]:
            c[o[0]] = c.get(o[0], 0) + 1
        counts[phase] = c
    doc = {
        "generator": "tests/golden/gen_call_trace.py",
        "backend": tb.__inner__,
        "what": "backend calls of one reference GenerateWhiskShuffleProof + IsValidWhiskShuffleProof at ell = 124 (seeds: CRS 9000, proof 9100)",
        "ell": ELL, "n_blinders": N_BLINDERS,
        "blob_sha256": hashlib.sha256(blob).hexdigest(), "blob_unit": 16,
        "op_format": {"gen": "[out]", "id": "[out]", "dec": "[out, enc_off, checked]", "add|sub": "[out, a, b]", "neg": "[out, a]", "mul": "[out, a, scalar_off]",
                      "eq": "[a, b, result]", "cmp": "[a, enc_off]", "msm": "[out, [bases], [scalar_offs]]", "acc_new": "[acc]",
                      "acc_check": "[acc, C, [bases], [scalar_offs], rho_off]", "acc_verify": "[acc, ok]"},
        "counts": counts,
        "verdict": bool(ok),
        "proof_sha256": hashlib.sha256(bytes(proof)).hexdigest(),
        "setup": ops[:n_setup], "prove": ops[n_setup:n_prove], "verify": ops[n_prove:],
    }
    with open(os.path.join(out_dir, "call_trace_ell124.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(json.dumps(counts, indent=1))
    print("blob", len(blob), "bytes; ops", len(ops), "; proof sha256", doc["proof_sha256"][:16])


if __name__ == "__main__":
    main()
