"""TEST INFRASTRUCTURE (CPU, big integers): the scalar side of the reference's shuffle VERIFIER, restated check by check.

Only tests/ may import this.  It follows /root/reference/curdleproofs/curdleproofs/ line by line -- the verification
scalars (util.py:71-78, ipa.py:164-186, same_msm.py:155-182), vec_u (grand_prod.py:147-156), inner_prod
(grand_prod.py:164-166), the polynomial factors (same_perm.py:98-101) and the eight `accumulate_check` calls plus the four
same-scalar equalities -- and answers one question: given the challenges the reference's transcript produced (recorded in
tests/golden/shuffle_vectors.json by the reference itself) and one weight per check, WHICH SCALAR does every point of the
statement carry in

    sum_k rho_k * ( C_k - MSM(bases_k, scalars_k) )  ==  identity        (msm_accumulator.py:37-68, with rho_k its random factor)

Every C_k is expanded into the wire / CRS points the reference builds it from with G1Point operators (e.g. point_lhs of
ipa.py:223, D of grand_prod.py:159, A' of curdleproofs.py:210).  Output rows use the reference's OWN orders: the instance
vec_R | vec_S | vec_T | vec_U (whisk_interface.py:96-100), then the proof's points in to_bytes() order (whisk_interface.py:58-61,
curdleproofs.py:275-285, same_perm.py:135-139, grand_prod.py:195-200, ipa.py:260-270, same_scalar.py:132-139,
same_msm.py:257-269); the CRS row in crs.py:92-101 order.  This is what the product's k_shuffle_rows / host front-end rows are
compared with (tests/test_shuffle_rows_oracle.py); nothing here imports the product.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001      # util.py:7 CURVE_ORDER
N_BLINDERS = 4                                                                   # curdleproofs.py:24


def inv(x: int) -> int:                                                          # util.py:51-54 invert
    return pow(x, -1, R)


def verification_scalars_bitstring(n: int, lg_n: int) -> List[List[int]]:        # util.py:71-78
    out = []
    for i in range(n):
        bs = bin(i)[2:].zfill(lg_n)
        out.append([j for j in range(lg_n) if bs[j] == "1"])
    return out


def vec_s_from(challenges: Sequence[int], n: int) -> List[int]:                  # ipa.py:179-183, same_msm.py:176-180
    bits = verification_scalars_bitstring(n, len(challenges))
    vec_s = []
    for i in range(n):
        s = 1
        for j in bits[i]:
            s = s * challenges[j] % R
        vec_s.append(s)
    return vec_s


def proof_fields(proof: bytes, ell: int) -> Dict[str, int]:
    """The Fr fields of a WhiskShuffleProof in wire form (BufReader.read_fr, util.py:149-153): offsets from the from_bytes chain."""
    n = ell + N_BLINDERS
    lg = n.bit_length() - 1
    off = 48 * 10                                       # M | A cm_T cm_U R S | B | C
    f = {}
    rd = lambda o: int.from_bytes(proof[o: o + 32], "little")
    f["r_p"] = rd(off); off += 32                        # grand_prod.py:211
    off += 48 * (2 + 4 * lg)                             # ipa.py:276-281
    f["c_final"] = rd(off); f["d_final"] = rd(off + 32); off += 64
    off += 48 * 4                                        # same_scalar.py:144-145
    f["z_k"], f["z_t"], f["z_u"] = rd(off), rd(off + 32), rd(off + 64); off += 96
    off += 48 * (3 + 6 * lg)                             # same_msm.py:274-282
    f["x_final"] = rd(off); off += 32
    assert off == len(proof), (off, len(proof))
    return f


def own_point_names(ell: int) -> List[Tuple]:
    """Instance, then the proof's points in wire order."""
    lg = (ell + N_BLINDERS).bit_length() - 1
    names: List[Tuple] = [(v, i) for v in ("vec_R", "vec_S", "vec_T", "vec_U") for i in range(ell)]
    names += [("M",), ("A",), ("cm_T.T_1",), ("cm_T.T_2",), ("cm_U.T_1",), ("cm_U.T_2",), ("R",), ("S",), ("B",), ("C",), ("B_c",), ("B_d",)]
    for v in ("vec_L_C", "vec_R_C", "vec_L_D", "vec_R_D"):
        names += [(v, j) for j in range(lg)]
    names += [("cm_A.T_1",), ("cm_A.T_2",), ("cm_B.T_1",), ("cm_B.T_2",), ("B_a",), ("B_t",), ("B_u",)]
    for v in ("vec_L_A", "vec_L_T", "vec_L_U", "vec_R_A", "vec_R_T", "vec_R_U"):
        names += [(v, j) for j in range(lg)]
    assert len(names) == 4 * ell + 19 + 10 * lg
    return names


def crs_point_names(ell: int) -> List[Tuple]:                                    # crs.py:92-101
    return [("crs.vec_G", i) for i in range(ell)] + [("crs.vec_H", i) for i in range(N_BLINDERS)] + \
           [("crs.H",), ("crs.G_t",), ("crs.G_u",), ("crs.G_sum",), ("crs.H_sum",)]


class _Statement:
    """sum of rho * (C - MSM(bases, scalars)) as a dictionary point-name -> scalar."""

    def __init__(self):
        self.coef: Dict[Tuple, int] = {}

    def add(self, name: Tuple, s: int) -> None:
        self.coef[name] = (self.coef.get(name, 0) + s) % R

    def check(self, rho: int, C: Dict[Tuple, int], bases: Sequence[Tuple], scalars: Sequence[int]) -> None:
        """msm_accumulator.py:37-58 with the random factor `rho`:  + rho * C  - rho * sum scalars[i] * bases[i]."""
        for name, s in C.items():
            self.add(name, rho * s)
        for b, s in zip(bases, scalars):
            if b is None:                               # an identity base (Z1): skipped, msm_accumulator.py:49-50
                continue
            self.add(b, -rho * s)


def _lin(*terms) -> Dict[Tuple, int]:
    """A point written as a linear combination of named points: _lin((name, scalar), ...)."""
    out: Dict[Tuple, int] = {}
    for name, s in terms:
        out[name] = (out.get(name, 0) + s) % R
    return out


def _scale(C: Dict[Tuple, int], k: int) -> Dict[Tuple, int]:
    return {n: s * k % R for n, s in C.items()}


def _plus(*Cs: Dict[Tuple, int]) -> Dict[Tuple, int]:
    out: Dict[Tuple, int] = {}
    for C in Cs:
        for n, s in C.items():
            out[n] = (out.get(n, 0) + s) % R
    return out


def statement_rows(ell: int, fields: Dict[str, int], challenges: Sequence[Tuple[str, int]], rho: Sequence[int]):
    """-> (own_row, crs_row): the scalar of every own point (own_point_names order) and of every CRS point (crs_point_names
    order).  `challenges`: (label, value) in the order the reference verifier drew them; `rho`: 12 weights -- rho[0..7] the
    random factors of the eight accumulate_check calls in call order, rho[8..11] the weights of the four same-scalar
    equalities (same_scalar.py:101-108: T_1 and T_2 components of expected_1 == computed_1, then of expected_2 == computed_2)."""
    n = ell + N_BLINDERS
    lg = n.bit_length() - 1
    ch = list(challenges)

    def draw(label: str) -> int:
        lab, v = ch.pop(0)
        assert lab == label, (lab, label)
        return v % R

    st = _Statement()
    G = [("crs.vec_G", i) for i in range(ell)]
    Hv = [("crs.vec_H", i) for i in range(N_BLINDERS)]

    # ---- curdleproofs.py:176-180
    vec_a = [draw("curdleproofs_vec_a") for _ in range(ell)]

    # ---- same_perm.py:91-109
    alpha, beta = draw("same_perm_alpha"), draw("same_perm_beta")
    gprod_result = 1
    for i, a in enumerate(vec_a):                                             # same_perm.py:98-101
        gprod_result = gprod_result * ((a + i * alpha + beta) % R) % R
    # accumulate_check((B - A) - M * alpha, crs_G_vec, [beta] * ell)           same_perm.py:103-107
    st.check(rho[0], _lin((("B",), 1), (("A",), -1), (("M",), -alpha)), G, [beta] * ell)

    # ---- grand_prod.py:137-166
    g_alpha = draw("gprod_alpha")
    g_beta = draw("gprod_beta")
    beta_inv = inv(g_beta)
    vec_u = []
    pw = beta_inv
    for _ in range(ell):                                                      # grand_prod.py:148-152
        vec_u.append(pw)
        pw = pw * beta_inv % R
    vec_u += [pow(beta_inv, ell + 1, R)] * N_BLINDERS                         # grand_prod.py:154
    D = _lin((("B",), 1), (("crs.G_sum",), -beta_inv), (("crs.H_sum",), g_alpha))   # grand_prod.py:157
    inner_prod = (fields["r_p"] * pow(g_beta, ell + 1, R) + gprod_result * pow(g_beta, ell, R) - 1) % R   # grand_prod.py:164-166
    vec_G = G + Hv                                                            # grand_prod.py:162

    # ---- ipa.py:204-236
    i_alpha, i_beta = draw("ipa_alpha"), draw("ipa_beta")
    gam = [draw("ipa_gamma") for _ in range(lg)]                              # ipa.py:168-176
    gam_inv = [inv(g) for g in gam]                                           # ipa.py:178
    vec_s = vec_s_from(gam, n)                                                # ipa.py:180-184
    vec_s_inv = [inv(s) for s in vec_s]                                       # ipa.py:186
    c, d = fields["c_final"], fields["d_final"]
    vec_rhs_scalars = [c * s % R for s in vec_s] + [c * d % R * i_beta % R]   # ipa.py:213-214
    vec_G_H = vec_G + [("crs.H",)]
    # H = crs_H * beta;  C_a = B_c + C * alpha + H * (alpha^2 inner_prod)       ipa.py:217-218
    C_a = _lin((("B_c",), 1), (("C",), i_alpha), (("crs.H",), i_beta * i_alpha % R * i_alpha % R * inner_prod))
    msm = lambda vec, sc: _lin(*[((vec, j), sc[j]) for j in range(lg)])
    st.check(rho[1], _plus(msm("vec_L_C", gam), C_a, msm("vec_R_C", gam_inv)), vec_G_H, vec_rhs_scalars)      # ipa.py:220-222
    vec_d_div_s = [d * (si * ui % R) % R for si, ui in zip(vec_s_inv, vec_u)]                                 # ipa.py:224-226
    D_a = _plus(_lin((("B_d",), 1)), _scale(D, i_alpha))                                                       # ipa.py:228
    st.check(rho[2], _plus(msm("vec_L_D", gam), D_a, msm("vec_R_D", gam_inv)), vec_G, vec_d_div_s)             # ipa.py:229-230

    # ---- same_scalar.py:82-108 (asserted exactly by the reference; weights rho[8..11] per component)
    s_alpha = draw("same_scalar_alpha")
    z_k, z_t, z_u = fields["z_k"], fields["z_t"], fields["z_u"]
    # expected_1 = (G_t z_t, R z_k + H z_t);  computed_1 = cm_A + cm_T * alpha       same_scalar.py:101-106, commitment.py:30
    st.check(rho[8], _lin((("crs.G_t",), z_t)), [("cm_A.T_1",), ("cm_T.T_1",)], [1, s_alpha])
    st.check(rho[9], _lin((("R",), z_k), (("crs.H",), z_t)), [("cm_A.T_2",), ("cm_T.T_2",)], [1, s_alpha])
    st.check(rho[10], _lin((("crs.G_u",), z_u)), [("cm_B.T_1",), ("cm_U.T_1",)], [1, s_alpha])
    st.check(rho[11], _lin((("S",), z_k), (("crs.H",), z_u)), [("cm_B.T_2",), ("cm_U.T_2",)], [1, s_alpha])

    # ---- curdleproofs.py:210-236 + same_msm.py:194-227
    A_prime = _lin((("A",), 1), (("cm_T.T_1",), 1), (("cm_U.T_1",), 1))                                        # curdleproofs.py:210
    vec_G_with_blinders = G + Hv[: N_BLINDERS - 2] + [("crs.G_t",), ("crs.G_u",)]                              # curdleproofs.py:212-214
    vec_T_with_blinders = [("vec_T", i) for i in range(ell)] + [None, None, ("crs.H",), None]                  # curdleproofs.py:216-221
    vec_U_with_blinders = [("vec_U", i) for i in range(ell)] + [None, None, None, ("crs.H",)]                  # curdleproofs.py:223-228
    m_alpha = draw("same_msm_alpha")
    gm = [draw("same_msm_gamma") for _ in range(lg)]                                                           # same_msm.py:158-173
    gm_inv = [inv(g) for g in gm]
    vec_sm = vec_s_from(gm, n)
    vec_x_times_s = [fields["x_final"] * s % R for s in vec_sm]                                                # same_msm.py:213
    A_a = _plus(_lin((("B_a",), 1)), _scale(A_prime, m_alpha))                                                 # same_msm.py:215
    Z_t_a = _lin((("B_t",), 1), (("cm_T.T_2",), m_alpha))                                                      # same_msm.py:216 (Z_t = cm_T.T_2)
    Z_u_a = _lin((("B_u",), 1), (("cm_U.T_2",), m_alpha))
    st.check(rho[3], _plus(msm("vec_L_A", gm), A_a, msm("vec_R_A", gm_inv)), vec_G_with_blinders, vec_x_times_s)   # same_msm.py:219-220
    st.check(rho[4], _plus(msm("vec_L_T", gm), Z_t_a, msm("vec_R_T", gm_inv)), vec_T_with_blinders, vec_x_times_s)  # same_msm.py:222-223
    st.check(rho[5], _plus(msm("vec_L_U", gm), Z_u_a, msm("vec_R_U", gm_inv)), vec_U_with_blinders, vec_x_times_s)  # same_msm.py:225-226

    # ---- curdleproofs.py:238-243
    st.check(rho[6], _lin((("R",), 1)), [("vec_R", i) for i in range(ell)], vec_a)
    st.check(rho[7], _lin((("S",), 1)), [("vec_S", i) for i in range(ell)], vec_a)
    assert not ch, "unused challenges: %r" % [c[0] for c in ch]

    own = [st.coef.pop(nm, 0) for nm in own_point_names(ell)]
    crs = [st.coef.pop(nm, 0) for nm in crs_point_names(ell)]
    assert not st.coef, "scalars on unknown points: %r" % list(st.coef)
    return own, crs, {"vec_s": vec_s, "vec_s_inv": vec_s_inv, "vec_sm": vec_sm, "vec_u": vec_u, "inner_prod": inner_prod,
                      "gprod_result": gprod_result, "beta_inv": beta_inv}
