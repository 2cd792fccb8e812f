"""Loader for tests/golden/shuffle_batch_ell124.{bin,json} (made by gen_shuffle_batch.py with the reference prover):
distinct ell = 124 Whisk shuffle proofs over one CRS + tampered variants with the reference verifier's verdicts, and the
extension shuffle_batch_ell124_more.{bin,json} (proofs 64 .. 1023 of the same seeded sequence: 1024 distinct proofs in all).
Data only."""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("pre_r", "pre_k", "post_r", "post_k", "proof")


class ShuffleBatch:
    def __init__(self):
        self.meta = json.load(open(os.path.join(HERE, "shuffle_batch_ell124.json")))
        blob = open(os.path.join(HERE, "shuffle_batch_ell124.bin"), "rb").read()
        assert hashlib.sha256(blob).hexdigest() == self.meta["sha256"]
        m = self.meta
        self.ell, self.count = m["ell"], m["count"]
        self.crs = blob[: m["crs_bytes"]]
        tb, pb, rb = m["tracker_bytes"], m["proof_bytes"], m["record_bytes"]
        self.inst_bytes, self.proof_bytes = 4 * tb, pb
        self.instances, self.proofs = [], []
        for i in range(self.count):
            rec = blob[m["crs_bytes"] + i * rb: m["crs_bytes"] + (i + 1) * rb]
            self.instances.append(rec[: 4 * tb])                 # pre_r | pre_k | post_r | post_k  = vec_R | vec_S | vec_T | vec_U
            self.proofs.append(rec[4 * tb:])
        more = os.path.join(HERE, "shuffle_batch_ell124_more.json")
        self.backends = {m["backend"]: self.count}
        if os.path.exists(more):
            mm = json.load(open(more))
            blob2 = open(os.path.join(HERE, "shuffle_batch_ell124_more.bin"), "rb").read()
            assert hashlib.sha256(blob2).hexdigest() == mm["sha256"] and mm["first"] == self.count and mm["record_bytes"] == rb
            assert mm["ell"] == self.ell and mm["crs_seed"] == m["crs_seed"] and mm["proof_seed_base"] == m["proof_seed_base"]
            for i in range(mm["count"]):
                rec = blob2[i * rb: (i + 1) * rb]
                self.instances.append(rec[: 4 * tb])
                self.proofs.append(rec[4 * tb:])
            self.count += mm["count"]
            self.backends[mm["backend"]] = self.backends.get(mm["backend"], 0) + mm["count"]
        self.tampered = []
        for t in m["tampered"]:
            bufs = dict(zip(KEYS, self._split(t["base"])))
            for which, off, hexbytes in t["edits"]:
                b = bytes.fromhex(hexbytes)
                bufs[which] = bufs[which][:off] + b + bufs[which][off + len(b):]
            self.tampered.append({"name": t["name"], "base": t["base"], "accepts": t["accepts"],
                                  "instance": bufs["pre_r"] + bufs["pre_k"] + bufs["post_r"] + bufs["post_k"], "proof": bufs["proof"]})

    def _split(self, i):
        tb = self.meta["tracker_bytes"]
        inst = self.instances[i]
        return [inst[k * tb: (k + 1) * tb] for k in range(4)] + [self.proofs[i]]

    def tiled(self, n, tampered_slots=None):
        """n proofs: the distinct ones cycled; tampered_slots = {slot: index into self.tampered} replaces some.
        -> (instances bytes, proofs bytes, expected verdict list)."""
        inst, prf, want = [], [], []
        tampered_slots = tampered_slots or {}
        for s in range(n):
            if s in tampered_slots:
                t = self.tampered[tampered_slots[s]]
                inst.append(t["instance"]); prf.append(t["proof"]); want.append(t["accepts"])
            else:
                inst.append(self.instances[s % self.count]); prf.append(self.proofs[s % self.count]); want.append(True)
        return b"".join(inst), b"".join(prf), want
