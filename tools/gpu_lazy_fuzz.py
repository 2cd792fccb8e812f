"""Differential campaign for the deferred G1Point layer ON THE GPU BOX: random programs over the operators, unchecked decodings, compute_MSM
(1 ... 96 terms, bases partly deferred values themselves) and MSMAccumulator sequences -- with points outside the prime-order subgroup
mixed in -- run on the product (operators deferred) and on the CPU oracle's value classes (oracle/py_arkworks_shim.py) side by side;
every byte string, comparison and verdict must agree.  The same programs then run with the operators computing at once.

    python tools/gpu_lazy_fuzz.py [seed] [programs]            -> profiles/r05_lazy_fuzz.txt
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import bls12_381 as O           # noqa: E402  (tools may use the oracle as a checker)
from oracle import py_arkworks_shim as S    # noqa: E402

T3 = (0, 2)


def oracle_msm(bases, scalars):
    cur = S.G1Point.identity()
    for b, s in zip(bases, scalars):
        cur = cur + b * s
    return cur


class OracleAccumulator:                      # msm_accumulator.py:32-68 over the oracle's values
    def __init__(self):
        self.A_c = S.G1Point.identity()
        self.map = {}

    def accumulate_check(self, C, bases, scalars, rho):
        self.A_c = self.A_c + C * rho
        for b, s in zip(bases, scalars):
            if b == S.G1Point.identity():
                continue
            k = bytes(b.to_compressed_bytes())
            self.map[k] = self.map.get(k, S.Scalar(0)) + rho * s

    def verify(self):
        if not self.map:
            raise ValueError("empty")
        keys, vals = zip(*self.map.items())
        return oracle_msm([S.G1Point.from_compressed_bytes_unchecked(k) for k in keys], vals) == self.A_c


def run_program(A, M, B, rng, n_ops, counts):
    mine, ref = [B.G1Point(), B.G1Point.identity()], [S.G1Point(), S.G1Point.identity()]
    seeds = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(5)]
    if rng.random() < 0.5:
        seeds += [O.g1_add(seeds[0], T3), O.g1_add(O.g1_mul(O.G1_GEN, 77), O.g1_neg(T3))]
        if rng.random() < 0.3:
            seeds.append(T3)
    for p in seeds:
        e = O.g1_compress(p)
        mine.append(B.G1Point.from_compressed_bytes_unchecked(e)); ref.append(S.G1Point.from_compressed_bytes_unchecked(e))
    for step in range(n_ops):
        op = rng.choice(["add", "add", "sub", "neg", "mul", "mul", "mul", "cmp", "eq", "dec", "msm", "msm", "acc"])
        i, j = rng.randrange(len(mine)), rng.randrange(len(mine))
        counts[op] = counts.get(op, 0) + 1
        if op == "add":
            mine.append(mine[i] + mine[j]); ref.append(ref[i] + ref[j])
        elif op == "sub":
            mine.append(mine[i] - mine[j]); ref.append(ref[i] - ref[j])
        elif op == "neg":
            mine.append(-mine[i]); ref.append(-ref[i])
        elif op == "mul":
            k = rng.choice([0, 1, 2, O.R - 1, rng.randrange(O.R), rng.randrange(O.R)])
            mine.append(mine[i] * B.Scalar(k)); ref.append(ref[i] * S.Scalar(k))
        elif op == "cmp":
            assert bytes(mine[i].to_compressed_bytes()) == bytes(ref[i].to_compressed_bytes()), ("cmp", step)
        elif op == "eq":
            assert (mine[i] == mine[j]) == (ref[i] == ref[j]), ("eq", step)
        elif op == "dec":
            e = bytes(ref[i].to_compressed_bytes())
            mine.append(B.G1Point.from_compressed_bytes_unchecked(e)); ref.append(S.G1Point.from_compressed_bytes_unchecked(e))
        elif op == "msm":
            n = rng.choice([1, 2, 3, 5, 8, 17, 40, 96])
            idx = [rng.randrange(len(mine)) for _ in range(n)]
            sc = [rng.choice([0, 1, rng.randrange(O.R), rng.randrange(O.R)]) for _ in range(n)]
            mine.append(A.compute_MSM([mine[t] for t in idx], [B.Scalar(s) for s in sc]))
            ref.append(oracle_msm([ref[t] for t in idx], [S.Scalar(s) for s in sc]))
        elif op == "acc":
            acc, oacc = A.MSMAccumulator(), OracleAccumulator()
            honest = rng.random() < 0.7
            for _ in range(rng.choice([1, 2, 3])):
                n = rng.choice([1, 4, 9])
                idx = [rng.randrange(len(mine)) for _ in range(n)]
                sc = [rng.randrange(O.R) for _ in range(n)]
                C_o = oracle_msm([ref[t] for t in idx], [S.Scalar(s) for s in sc])
                C_m = A.compute_MSM([mine[t] for t in idx], [B.Scalar(s) for s in sc])
                if not honest:
                    C_o = C_o + ref[0]; C_m = C_m + mine[0]
                rho = rng.randrange(1, O.R)
                orig = M.random_scalar
                M.random_scalar = lambda rho=rho: B.Scalar(rho)
                try:
                    acc.accumulate_check(C_m, [mine[t] for t in idx], [B.Scalar(s) for s in sc])
                finally:
                    M.random_scalar = orig
                oacc.accumulate_check(C_o, [ref[t] for t in idx], [S.Scalar(s) for s in sc], S.Scalar(rho))
            try:
                want = oacc.verify()
            except ValueError:
                want = None
            try:
                acc.verify(); got = True
            except AssertionError:
                got = False
            except ValueError:
                got = None
            assert got == want, ("acc", step, got, want)
    for a, b in zip(mine, ref):
        assert str(a) == str(b)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    programs = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    from curdleproofs_pie_amd import _native as N

    N.tune_runtime()
    import curdleproofs_pie_amd as A
    import curdleproofs_pie_amd.msm_accumulator as M
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B

    t0 = time.perf_counter()
    for lazy in (True, False):
        B.set_lazy(lazy)
        counts = {}
        s0 = dict(B.stats)
        for p in range(programs):
            run_program(A, M, B, random.Random(seed * 100003 + p), 45, counts)
            if (p + 1) % 20 == 0:
                print(f"{'deferred' if lazy else 'eager'}: {p + 1} programs ok ({time.perf_counter() - t0:.0f} s)", flush=True)
        d = {k: B.stats[k] - s0[k] for k in B.stats}
        print(f"{'deferred' if lazy else 'eager'}: {programs} programs x 45 operations agree with the oracle; operations {dict(sorted(counts.items()))}; evaluation {d}", flush=True)
    print(f"lazy fuzz ok: seed {seed}, {time.perf_counter() - t0:.0f} s")


if __name__ == "__main__":
    main()
