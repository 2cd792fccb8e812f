"""Prover-side fold / map kernels driven with protocol-shaped data: inputs and outputs of the reference
prover's hot loops recorded by tests/golden/gen_prover_golden.py (reference classes over the CPU oracle).  Every L / R point
of every halving round, the final scalars, vec_T / vec_U / M and the grand-product bases must come out byte-identical."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pv(native_lib):
    return json.load(open(os.path.join(ROOT, "tests", "golden", "prover_vectors.json")))


def P(h):
    from curdleproofs_pie_amd.py_arkworks_bls12381 import G1Point
    return G1Point.from_compressed_bytes_unchecked(bytes.fromhex(h))


def S(h):
    from curdleproofs_pie_amd.py_arkworks_bls12381 import Scalar
    return Scalar.from_le_bytes(bytes.fromhex(h))


def enc(points):
    return [bytes(p.to_compressed_bytes()).hex() for p in points]


def test_ipa_halving_rounds(pv):
    from curdleproofs_pie_amd.prover_kernels import ipa_rounds

    r = pv["ipa"]
    gammas = [S(g) for g in r["gammas"]]
    seen = []

    def next_gamma(L_C, L_D, R_C, R_D):
        seen.append(enc([L_C, L_D, R_C, R_D]))
        return gammas.pop(0)

    LC, RC, LD, RD, c_fin, d_fin = ipa_rounds([P(h) for h in r["crs_G_vec"]], [P(h) for h in r["crs_G_prime_vec"]], P(r["H"]),
                                              [S(h) for h in r["vec_c"]], [S(h) for h in r["vec_d"]], next_gamma)
    assert not gammas and len(seen) == 5                                    # log2(32) rounds
    assert (enc(LC), enc(RC), enc(LD), enc(RD)) == (r["vec_L_C"], r["vec_R_C"], r["vec_L_D"], r["vec_R_D"])
    assert bytes(c_fin.to_le_bytes()).hex() == r["c_final"] and bytes(d_fin.to_le_bytes()).hex() == r["d_final"]


def test_same_msm_halving_rounds(pv):
    from curdleproofs_pie_amd.prover_kernels import same_msm_rounds

    r = pv["same_msm"]
    gammas = [S(g) for g in r["gammas"]]
    out = same_msm_rounds([P(h) for h in r["crs_G_vec"]], [P(h) for h in r["vec_T"]], [P(h) for h in r["vec_U"]],
                          [S(h) for h in r["vec_x"]], lambda *pts: gammas.pop(0))
    assert not gammas
    for got, key in zip(out[:6], ("vec_L_A", "vec_L_T", "vec_L_U", "vec_R_A", "vec_R_T", "vec_R_U")):
        assert enc(got) == r[key], key
    assert bytes(out[6].to_le_bytes()).hex() == r["x_final"]


def test_permute_and_commit(pv, monkeypatch):
    import curdleproofs_pie_amd.prover_kernels as K

    r = pv["permute_commit"]
    blinders = [S(b) for b in r["blinders"]]
    monkeypatch.setattr(K, "random_scalar", lambda: blinders.pop(0))

    class Crs:
        pass

    crs = Crs()
    g = pv["grand_product_bases"]
    crs.vec_G, crs.vec_H = [P(h) for h in g["vec_G"]], [P(h) for h in g["vec_H"]]
    vec_T, vec_U, M, bl = K.shuffle_permute_and_commit_input(crs, [P(h) for h in r["vec_R"]], [P(h) for h in r["vec_S"]], r["permutation"], S(r["k"]))
    assert not blinders and [bytes(b.to_le_bytes()).hex() for b in bl] == r["blinders"]
    assert enc(vec_T) == r["vec_T"] and enc(vec_U) == r["vec_U"] and enc([M]) == [r["M"]]


def test_grand_product_bases(pv):
    from curdleproofs_pie_amd.prover_kernels import grand_product_bases

    g = pv["grand_product_bases"]
    Gp, Hp = grand_product_bases([P(h) for h in g["vec_G"]], [P(h) for h in g["vec_H"]], S(g["beta_inv"]))
    assert enc(Gp + Hp) == g["G_prime_H_prime"]


def test_cross_proof_batched_rounds_equal_single_prover_rounds(pv):
    """ipa_rounds_many / same_msm_rounds_many: three provers in step (the reference-recorded inputs, and the same inputs with the
    vectors of prover 2 and 3 rotated so the three differ) -- prover 0's outputs must still be the reference's bytes, and every
    prover's outputs must equal what it gets alone."""
    from curdleproofs_pie_amd.prover_kernels import ipa_rounds, ipa_rounds_many, same_msm_rounds, same_msm_rounds_many

    r = pv["ipa"]
    rot = lambda v, k: v[k:] + v[:k]
    base = ([P(h) for h in r["crs_G_vec"]], [P(h) for h in r["crs_G_prime_vec"]], P(r["H"]), [S(h) for h in r["vec_c"]], [S(h) for h in r["vec_d"]])
    provers = [base] + [(base[0], base[1], base[2], rot(base[3], k), rot(base[4], 2 * k)) for k in (1, 2)]
    mk = lambda: (lambda gs: (lambda *pts: gs.pop(0)))([S(g) for g in r["gammas"]])
    many = ipa_rounds_many(provers, [mk() for _ in provers])
    assert (enc(many[0][0]), enc(many[0][1]), enc(many[0][2]), enc(many[0][3])) == (r["vec_L_C"], r["vec_R_C"], r["vec_L_D"], r["vec_R_D"])
    for pr, got in zip(provers, many):
        alone = ipa_rounds(*pr, mk())
        assert [enc(x) for x in got[:4]] == [enc(x) for x in alone[:4]] and got[4] == alone[4] and got[5] == alone[5]
    r = pv["same_msm"]
    base = ([P(h) for h in r["crs_G_vec"]], [P(h) for h in r["vec_T"]], [P(h) for h in r["vec_U"]], [S(h) for h in r["vec_x"]])
    provers = [base] + [(base[0], rot(base[1], k), base[2], rot(base[3], k)) for k in (3, 5)]
    mk = lambda: (lambda gs: (lambda *pts: gs.pop(0)))([S(g) for g in r["gammas"]])
    many = same_msm_rounds_many(provers, [mk() for _ in provers])
    for got, key in zip(many[0][:6], ("vec_L_A", "vec_L_T", "vec_L_U", "vec_R_A", "vec_R_T", "vec_R_U")):
        assert enc(got) == r[key], key
    for pr, got in zip(provers, many):
        alone = same_msm_rounds(*pr, mk())
        assert [enc(x) for x in got[:6]] == [enc(x) for x in alone[:6]] and got[6] == alone[6]


def test_implicit_base_change_equals_explicit_bases(pv):
    """ipa_rounds with G' given implicitly (the CRS points + the coefficients beta^-(i+1) of grand_prod.py:64-71) produces the very
    L / R points and final scalars it produces over the materialised G' -- the base change then costs no scalar multiplication."""
    from curdleproofs_pie_amd.prover_kernels import grand_product_bases, grand_product_coeffs, ipa_rounds

    g, r = pv["grand_product_bases"], pv["ipa"]
    G, H4 = [P(h) for h in g["vec_G"]], [P(h) for h in g["vec_H"]]
    n = len(r["vec_c"])
    assert len(G) + len(H4) >= n
    bases = (G + H4)[:n]
    beta_inv = S(g["beta_inv"])
    coeffs = grand_product_coeffs(len(G), len(H4), beta_inv)[:n]
    Gp, Hp = grand_product_bases(G, H4, beta_inv)
    explicit = (Gp + Hp)[:n]
    mk = lambda: (lambda gs: (lambda *pts: gs.pop(0)))([S(x) for x in r["gammas"]])
    c, d, Hh = [S(h) for h in r["vec_c"]], [S(h) for h in r["vec_d"]], P(r["H"])
    a = ipa_rounds(bases, explicit, Hh, c, d, mk())
    b = ipa_rounds(bases, bases, Hh, c, d, mk(), G_prime_coeffs=coeffs)
    assert [enc(x) for x in a[:4]] == [enc(x) for x in b[:4]] and a[4] == b[4] and a[5] == b[5]
