import json, os, random, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from curdleproofs_pie_amd import _native as N
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
from test_shuffle_verifier import apply_edits
from oracle import c_oracle
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))
ctx = N.default_context()
for mode in ("merged", "independent"):
    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx)
        got = v.verify_many([apply_edits(case, x["edits"]) for x in case["variants"]], mode=mode, rng=random.Random(case["seed"]))
        print(mode, case["ell"], got == [x["accepts"] for x in case["variants"]], v.last_stats.get("merged_ok"))
case = gold["cases"][4]
v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx)
item = apply_edits(case, [])
bad = apply_edits(case, case["variants"][0]["edits"])
got = v.verify_many([item] * 64, rng=random.Random(1))
print("64 valid:", all(got), v.last_stats["merged_ok"])
n = 64; L, C = v.crs.points_per_proof, v.crs.ncrs
b = v._bufs[n]; tot = n * L + C
pts = b["pts"].download(tot * 96); sc = b["sc"].download(tot * 32)
print("64: oracle identity:", c_oracle.msm_bucket(pts, sc, tot) == bytes(96))
batch = [item] * 20 + [bad] + [item] * 11
got = v.verify_many(batch, mode="merged", rng=random.Random(2))
print("merged_ok", v.last_stats.get("merged_ok"), "status", v.last_status)
n = len(batch)
b = v._bufs[n]
tot = n * L + C
pts = b["pts"].download(tot * 96); sc = b["sc"].download(tot * 32)
for c in (0, 8, 16):
    blob = ctx.msm_device(b["pts"], b["sc"], tot, window_c=c)
    print("window", c, "identity:", bool(N.cg1_is_identity(blob)))
o = c_oracle.msm_bucket(pts, sc, tot)
print("oracle identity:", o == bytes(96))
i = 20
print("bad proof scalars all zero:", sc[i*L*32:(i+1)*L*32] == bytes(L*32))
print("crs part equals host crs:", pts[n*L*96:] == v.crs.affine96)
