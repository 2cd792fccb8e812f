#!/usr/bin/env python3
"""Randomised differential campaign for cg1_merlin_batch_device (not part of the test-suite): random operation lists (the generator of
tests/test_merlin_block_program.py) on 70 lanes with different data, through the block program (when the list fits), the byte-level
machine and the one-lane-at-a-time kernel; outputs and final 208-byte states must equal the host transcript's, lane by lane.

    python tools/gpu_merlin_fuzz.py SEED SECONDS
"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from curdleproofs_pie_amd import _native as N
import curdleproofs_pie_amd.merlin as M
from test_merlin_block_program import random_program, host_run_with_prog_label

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
ctx = N.default_context()
rng = random.Random(seed0)
t0 = time.time(); it = 0; used = {0: 0, 1: 0, 2: 0}
while time.time() - t0 < budget:
    it += 1
    prog, plan, nbytes = random_program(M, random.Random(rng.getrandbits(32)), rng.randrange(1, 80))
    n = rng.choice([1, 3, 64, 70])
    rows = [bytes(rng.randrange(256) for _ in range(max(1, nbytes))) for _ in range(n)]
    want = [host_run_with_prog_label(M, prog, plan, r) for r in rows[:4]] + [host_run_with_prog_label(M, prog, plan, rows[-1])]
    for rows_form, sync in ((1, 1), (0, 1), (0, 0)):
        ctx.set_param("merlin_rows", rows_form); ctx.set_param("merlin_sync", sync)
        outs, states = prog.run(rows, ctx, want_states=True)
        used[N.cg1_merlin_last_kernel(ctx.handle)] += 1
        for i, (w_out, w_state) in zip(list(range(min(4, n))) + [n - 1], want):
            for o, v in w_out.items():
                if outs[i][o: o + len(v)] != v:
                    print("MISMATCH output", dict(seed=seed0, it=it, kernel=(rows_form, sync), lane=i, off=o), flush=True); sys.exit(1)
            if states[i][:203] != w_state:
                print("MISMATCH state", dict(seed=seed0, it=it, kernel=(rows_form, sync), lane=i), flush=True); sys.exit(1)
    if it % 50 == 0:
        print(f"{it} programs ok ({time.time() - t0:.0f} s); calls served by block program / byte machine / one lane at a time: {used[2]} / {used[1]} / {used[0]}", flush=True)
ctx.set_param("merlin_rows", 1); ctx.set_param("merlin_sync", 1)
print(f"merlin fuzz ok: {it} programs, seed {seed0}; calls served by block program / byte machine / one lane at a time: {used[2]} / {used[1]} / {used[0]}")
