#!/usr/bin/env python3
"""Throughput / latency of the batched device transcript (k_merlin_batch) on a program with the SHAPE of one ell = 124 shuffle
verification (curdleproofs.py:162-248 and its sub-arguments: 4 ell + 1 points, ell challenges, ell + 2 appended scalars,
..., 974 appended messages and 146 rejection-sampled challenges), random data per lane, against the host transcript
(one thread; the grouped AVX-512 front-end is ~3x faster per core than this plain loop)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N
import curdleproofs_pie_amd.merlin as M

ell, lg = 124, 7
prog = M.TranscriptProgram(b"curdleproofs")
off = [0]
plan = []
def pt(label):
    prog.append(label, off[0], 48); plan.append((label, off[0], 48)); off[0] += 48
def sc(label):
    prog.append(label, off[0], 32); plan.append((label, off[0], 32)); off[0] += 32
def ch(label):
    plan.append((label, prog.challenge_scalar(label), None))
for _ in range(4 * ell + 1): pt(b"curdleproofs_step1")
for _ in range(ell): ch(b"curdleproofs_vec_a")
pt(b"same_perm_step1"); pt(b"same_perm_step1")
for _ in range(ell): sc(b"same_perm_step1")
ch(b"same_perm_alpha"); ch(b"same_perm_beta")
pt(b"gprod_step1"); sc(b"gprod_step1"); ch(b"gprod_alpha"); pt(b"gprod_step2"); sc(b"gprod_step2"); ch(b"gprod_beta")
pt(b"ipa_step1"); pt(b"ipa_step1"); sc(b"ipa_step1"); pt(b"ipa_step1"); pt(b"ipa_step1"); ch(b"ipa_alpha"); ch(b"ipa_beta")
for _ in range(lg):
    for _ in range(4): pt(b"ipa_loop")
    ch(b"ipa_gamma")
for _ in range(10): pt(b"sameexp_points")
ch(b"same_scalar_alpha")
for _ in range(3 + 2 * (ell + 4) + 3): pt(b"same_msm_step1")
ch(b"same_msm_alpha")
for _ in range(lg):
    for _ in range(6): pt(b"same_msm_loop")
    ch(b"same_msm_gamma")
nap = sum(1 for p in plan if p[2] is not None); nch = len(plan) - nap
print(f"program: {nap} appended messages, {nch} challenges, {off[0]} data bytes per transcript", flush=True)
ctx = N.Context(0)
rng = random.Random(1)
base = bytes(rng.randrange(256) for _ in range(off[0]))
for rows_form, sync, lanes in ((0, 0, 64), (0, 1, 64), (1, 1, 64), (1, 1, 16), (0, 1, 16), (0, 1, 1)):
    ctx.set_param("merlin_rows", rows_form)
    ctx.set_param("merlin_sync", sync)
    ctx.set_param("merlin_lanes", lanes)
    print((f"k_merlin_batch_rows (block program: whole rate blocks per pass), {lanes} transcripts per wave" if rows_form else
           f"k_merlin_batch_sync (byte-level state machine, lanes of a wave permute together), {lanes} transcripts per wave") if sync
          else "k_merlin_batch (round 2: every lane permutes where its own transcript needs it)", flush=True)
    for n in ((64, 1024, 4096, 16384) if lanes == 64 else (1024, 4096)):
        rows = [base[i % 97:] + base[: i % 97] for i in range(n)]
        prog.run(rows[:64], ctx)
        t0 = time.perf_counter(); outs, _ = prog.run(rows, ctx); dt = time.perf_counter() - t0
        print(f"  n={n}: device call {prog.last_call_ms:.2f} ms ({n/prog.last_call_ms*1e3:.0f} transcripts/s), Keccak passes of the slowest wave {prog.last_passes}; "
              f"{1e3*dt:.1f} ms wall incl. H2D of {n*off[0]/1e6:.0f} MB and Python packing", flush=True)
    assert N.cg1_merlin_last_kernel(ctx.handle) == (2 if rows_form else sync)
    if sync == 0:
        ref_outs = outs
    else:
        assert outs == ref_outs[: len(outs)], "the two kernels disagree"
t = M.CurdleproofsTranscript(b"curdleproofs")
t0 = time.perf_counter()
for label, o, ln in plan:
    if ln is None:
        got = bytes(t.get_and_append_challenge(label).to_le_bytes()); assert got == outs[0][o: o + 32]
    else:
        t.append(label, rows[0][o: o + ln])
print(f"host transcript (ctypes call per operation, one thread): {1e3*(time.perf_counter()-t0):.2f} ms; challenges equal lane 0's", flush=True)
