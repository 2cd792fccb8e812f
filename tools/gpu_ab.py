#!/usr/bin/env python3
"""A/B two builds of the library on one box: alternate `bench.py --no-secondary` runs and print k_accumulate / step times.
usage: python tools/gpu_ab.py LIB_A LIB_B [rounds]   ("-" = the default library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
extra = sys.argv[4:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "-":
            env["CURDLE_G1_LIB"] = os.path.abspath(l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-secondary", "--no-cpu-baseline", "--steps", "20", "--warmup", "5", *extra], env=env, capture_output=True, text=True)
        if out.returncode != 0 or not out.stdout.strip():
            sys.exit(f"bench.py failed with {l}: rc={out.returncode}\n{out.stderr[-2000:]}")
        out = out.stdout
        d = json.loads(out.strip().splitlines()[-1])
        res[l].append((d["ms_per_step"], d["roofline"]["kernel_ms"]))
        print(r, l, "step %.3f ms  k_accumulate %.3f ms" % res[l][-1], flush=True)
for l in libs:
    s = sorted(x[0] for x in res[l]); k = sorted(x[1] for x in res[l])
    print(f"{l}: step median {s[len(s)//2]:.3f} min {s[0]:.3f} | k_accumulate median {k[len(k)//2]:.3f} min {k[0]:.3f}")
