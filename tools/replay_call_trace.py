"""Replay the reference's own backend-call sequence (tests/golden/call_trace_ell124.{json,bin}: one GenerateWhiskShuffleProof and one
IsValidWhiskShuffleProof at ell = 124, recorded by tests/golden/gen_call_trace.py over the oracle backend) through the product's
Python face.  Every output the reference's caller could observe -- compressed bytes, equality results, the verifier's verdict --
must come back bit for bit; the wall time is what the reference's UNCHANGED control flow costs on this backend (its scalar
arithmetic and transcript, which are the caller's own Python, are not in the trace).

    python tools/replay_call_trace.py [--reps 5]            -> profiles/r05_call_trace_replay.txt
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def load(name="call_trace_ell124"):
    import hashlib

    with open(os.path.join(GOLD, name + ".json")) as f:
        doc = json.load(f)
    with open(os.path.join(GOLD, name + ".bin"), "rb") as f:
        blob = f.read()
    assert hashlib.sha256(blob).hexdigest() == doc["blob_sha256"], "call-trace blob does not match its header"
    return doc, blob


class Replayer:
    """Executes op lists over a backend: G1Point / Scalar classes, compute_MSM(bases, scalars), an MSMAccumulator class whose
    accumulate_check draws its random factor through `set_rho` (so the recorded draw can be replayed)."""

    def __init__(self, doc, blob, G1Point, Scalar, compute_MSM, MSMAccumulator, set_rho):
        self.doc, self.blob = doc, blob
        self.G1Point, self.Scalar, self.compute_MSM, self.MSMAccumulator, self.set_rho = G1Point, Scalar, compute_MSM, MSMAccumulator, set_rho
        self.vals = {}
        self.accs = {}
        self.mismatches = []
        self._sc_cache = {}

    def scalar(self, off):
        s = self._sc_cache.get(off)
        if s is None:
            s = self._sc_cache[off] = self.Scalar(int.from_bytes(self.blob[16 * off: 16 * off + 32], "little"))
        return s

    def prepare(self, ops):
        """The Scalar objects of a phase exist before the calls are made (the callers computed them): build them outside the clock."""
        for o in ops:
            k = o[0]
            if k == "mul":
                self.scalar(o[3])
            elif k == "msm":
                for off in o[3]:
                    self.scalar(off)
            elif k == "acc_check":
                for off in o[4]:
                    self.scalar(off)
                self.scalar(o[5])

    def run(self, ops, by_kind=None):
        V, B, G = self.vals, self.blob, self.G1Point
        sc = self._sc_cache
        t_phase = time.perf_counter()
        for o in ops:
            k = o[0]
            t0 = time.perf_counter()
            if k == "add":
                V[o[1]] = V[o[2]] + V[o[3]]
            elif k == "mul":
                V[o[1]] = V[o[2]] * sc[o[3]]
            elif k == "cmp":
                if bytes(V[o[1]].to_compressed_bytes()) != B[16 * o[2]: 16 * o[2] + 48]:
                    self.mismatches.append(o)
            elif k == "dec":
                enc = B[16 * o[2]: 16 * o[2] + 48]
                V[o[1]] = G.from_compressed_bytes(enc) if o[3] else G.from_compressed_bytes_unchecked(enc)
            elif k == "msm":
                V[o[1]] = self.compute_MSM([V[i] for i in o[2]], [sc[j] for j in o[3]])
            elif k == "sub":
                V[o[1]] = V[o[2]] - V[o[3]]
            elif k == "neg":
                V[o[1]] = -V[o[2]]
            elif k == "eq":
                if (V[o[1]] == V[o[2]]) != o[3]:
                    self.mismatches.append(o)
            elif k == "acc_check":
                self.set_rho(sc[o[5]])
                self.accs[o[1]].accumulate_check(V[o[2]], [V[i] for i in o[3]], [sc[j] for j in o[4]])
            elif k == "acc_verify":
                try:
                    self.accs[o[1]].verify()
                    ok = True
                except AssertionError:
                    ok = False
                if ok != o[2]:
                    self.mismatches.append(o)
            elif k == "acc_new":
                self.accs[o[1]] = self.MSMAccumulator()
            elif k == "gen":
                V[o[1]] = G()
            elif k == "id":
                V[o[1]] = G.identity()
            else:
                raise ValueError("unknown op %r" % (k,))
            if by_kind is not None:
                e = by_kind.setdefault(k, [0, 0.0])
                e[0] += 1
                e[1] += time.perf_counter() - t0
        return time.perf_counter() - t_phase


def product_replayer(doc, blob):
    import curdleproofs_pie_amd as A
    import curdleproofs_pie_amd.msm_accumulator as M

    box = {}

    def set_rho(s):
        box["rho"] = s

    class ReplayAccumulator(A.MSMAccumulator):
        """msm_accumulator.py:43 draws one random factor per accumulate_check: here the recorded one (the module's random_scalar is swapped
        for the duration of the call only)."""

        def accumulate_check(self, C, bases, scalars):
            orig = M.random_scalar
            M.random_scalar = lambda: box.pop("rho")
            try:
                super().accumulate_check(C, bases, scalars)
            finally:
                M.random_scalar = orig

    return Replayer(doc, blob, A.G1Point, A.Scalar, A.compute_MSM, ReplayAccumulator, set_rho)


def _measure_mode(doc, blob, reps):
    best = {}
    parity = True
    for r in range(reps):
        rp = product_replayer(doc, blob)
        for ph in ("setup", "prove", "verify"):
            rp.prepare(doc[ph])
        times, kinds = {}, {}
        for ph in ("setup", "prove", "verify"):
            kinds[ph] = {}
            times[ph] = rp.run(doc[ph], kinds[ph])
        parity = parity and not rp.mismatches
        for ph in ("prove", "verify"):
            if ph not in best or times[ph] < best[ph][0]:
                best[ph] = (times[ph], kinds[ph])
    return best, parity


def measure(reps=3, ab=True):
    """{"verify_ms", "prove_ms", by-kind splits, "parity"}: best of `reps` full replays (fresh value table each time; the normal-form and
    resident-vector caches of the Python face behave as in a long-lived process: CRS points stay the same objects across proofs) with
    the operators deferred (the default), and -- `ab` -- the same with every operator computing at once (CURDLE_G1_LAZY=0, round 4)."""
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B

    doc, blob = load()
    out = {"what": "the reference's backend calls of one ell = 124 proof replayed through curdleproofs_pie_amd, its control flow unchanged: the G1Point operators return "
                   "deferred values, evaluated in batches (host worker pool for a handful of operator results, the GPU's batched MSM for the rest and for every compute_MSM) "
                   "when bytes or a comparison are asked for; compute_MSM / MSMAccumulator on the GPU",
           "ops": {ph: sum(doc["counts"][ph].values()) for ph in ("setup", "prove", "verify")}}
    prev = B.set_lazy(True)
    try:
        s0 = dict(B.stats)
        best, parity = _measure_mode(doc, blob, reps)
        s1 = dict(B.stats)
        out["deferred"] = True
        for ph in ("prove", "verify"):
            out[ph + "_ms"] = best[ph][0] * 1e3
            out[ph + "_by_kind_ms"] = {k: {"calls": v[0], "ms": round(v[1] * 1e3, 3)} for k, v in sorted(best[ph][1].items(), key=lambda kv: -kv[1][1])}
        out["per_replay"] = {k: round((s1[k] - s0[k]) / reps, 1) for k in s1}
        if ab:
            B.set_lazy(False)
            eb, ep = _measure_mode(doc, blob, max(1, reps - 1))
            out["eager"] = {"verify_ms": eb["verify"][0] * 1e3, "prove_ms": eb["prove"][0] * 1e3, "parity": ep}
            parity = parity and ep
    finally:
        B.set_lazy(prev)
    out["parity"] = parity
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    print(json.dumps(measure(a.reps), indent=1))
