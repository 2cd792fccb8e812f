#!/usr/bin/env python3
"""Generate tests/golden/msm_vectors.json with the CPU oracle (oracle/bls12_381.py, cross-checked against
oracle/msm_oracle.c).  The reference itself holds NO golden MSM outputs: its only
curve-level known answers are the generator and 99*G, which pin the oracle (tests/test_oracle_kat.py).
These vectors are therefore oracle-generated regression fixtures -- inputs as 48-byte compressed points and
32-byte LE scalars (hex), expected output as the 48-byte compression `G1Point.to_compressed_bytes()` returns.

    python tests/golden/gen_golden.py      # rewrites msm_vectors.json deterministically (seeded)
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bls12_381 as O  # noqa: E402
from oracle import c_oracle as C  # noqa: E402


def raw96(pt):
    return bytes(96) if pt is None else pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little")


def case(name, pts, scalars):
    want = O.compute_MSM(pts, scalars) if len(pts) <= 64 else O.compute_MSM_fast(pts, scalars)
    got_c = C.compute_msm(b"".join(raw96(p) for p in pts), b"".join((s % O.R).to_bytes(32, "little") for s in scalars), len(pts))
    assert got_c == raw96(want), name
    return {
        "name": name,
        "points": [O.g1_compress(p).hex() for p in pts],
        "scalars": [(s % O.R).to_bytes(32, "little").hex() for s in scalars],
        "expected": O.g1_compress(want).hex(),
    }


def main():
    rng = random.Random(20241008)
    pool = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(640)]
    rs = lambda: rng.randint(1, O.R - 1)  # the reference's random_scalar distribution (util.py:21-24)
    cases = []
    # k * G known answers (k small and k = r - 1)
    for k in (1, 2, 3, 99, O.R - 1):
        cases.append(case(f"kG_{k if k < 1000 else 'r_minus_1'}", [O.G1_GEN], [k]))
    for n in (1, 2, 3, 7, 64, 307, 627):   # 307 / 627 = final accumulator MSM at N=64 / N=128
        cases.append(case(f"random_n{n}", pool[:n], [rs() for _ in range(n)]))
    P, Q = pool[0], pool[1]
    cases.append(case("zero_scalar", [P, Q], [0, 5]))
    cases.append(case("all_zero_scalars", [P, Q], [0, 0]))
    cases.append(case("identity_base", [None, P, None], [7, 11, 13]))
    cases.append(case("duplicate_bases", [P, P, P, Q, Q], [3, 3, 5, 9, 9]))
    cases.append(case("p_and_minus_p_cancel", [P, O.g1_neg(P)], [12345, 12345]))
    cases.append(case("p_and_minus_p", [P, O.g1_neg(P), Q], [100, 99, 1]))
    beta = rs()
    cases.append(case("all_equal_scalars_n124", pool[:124], [beta] * 124))      # same_perm.py:54-55 pattern
    cases.append(case("sigma_scalars_0_to_123", pool[:124], list(range(124))))  # curdleproofs.py:315 pattern
    cases.append(case("max_scalars", pool[:5], [O.R - 1] * 5))
    cases.append(case("powers_of_two", pool[:16], [1 << (16 * i + 15) for i in range(16)]))  # window-boundary digits
    cases.append(case("all_ones_digits", pool[:4], [(1 << 255) % O.R - 1, 0x7FFF7FFF7FFF, (1 << 128) - 1, 0x8000]))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "msm_vectors.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/gen_golden.py", "seed": 20241008, "cases": cases}, f, indent=0)
    print("wrote", out, len(cases), "cases", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
