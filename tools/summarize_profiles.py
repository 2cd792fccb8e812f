#!/usr/bin/env python3
"""Turn gpurun_out/TAG_* (from tools/collect_profiles.sh) into committed summaries under profiles/.
usage: python tools/summarize_profiles.py TAG OUTPREFIX   e.g.  r01v4 profiles/r01_v4"""
import collections
import csv
import glob
import json
import shutil
import sys

tag, out = sys.argv[1], sys.argv[2]


def agg(pattern):
    files = glob.glob(pattern)
    a = collections.defaultdict(list)
    if not files:
        return a
    for r in csv.DictReader(open(files[0])):
        a[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in a.items()}


shutil.copy(glob.glob(f"gpurun_out/{tag}_kt/*/*_kernel_stats.csv")[0], out + "_kernel_stats.csv")
shutil.copy(f"gpurun_out/{tag}_bench.json", out + "_bench.json")
import subprocess
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import build as _B
os.makedirs("build/asm", exist_ok=True)
if _B.build_info().get("pipeline") == "staged":
    _B.emit_device_asm("build/asm/dev_current.s")
else:
    subprocess.check_call(["hipcc", "-S", "--offload-device-only", "--offload-arch=gfx950", "-O3", "-std=c++17", "curdleproofs_pie_amd/csrc/msm_gpu.hip", "-o", "build/asm/dev_current.s"])
res = subprocess.run([sys.executable, "tools/asm_stats.py", "build/asm/dev_current.s"], capture_output=True, text=True).stdout
open(out + "_kernel_resources.txt", "w").write("# per-kernel resource usage and instruction mix of the shipped device code (build.py emit_device_asm, the staged pipeline's own listing; tools/asm_stats.py)\n" + res)
f, w, q = agg(f"gpurun_out/{tag}_fetch/*/*_counter_collection.csv"), agg(f"gpurun_out/{tag}_write/*/*_counter_collection.csv"), agg(f"gpurun_out/{tag}_sq/*/*_counter_collection.csv")
lines = ["# rocprofv3 --pmc summaries of `python3 bench.py` (one 2^20-term MSM per step, c=16), MI355X",
         "# separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM_GUI_ACTIVE.  FETCH/WRITE in KiB per launch (average over launches).",
         "# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B read requests at 64 B -> doubled in hbm_bytes.",
         "kernel,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes=(2*F+W)*1024,launches"]
for k in sorted({k[0] for k in f}):
    F = f.get((k, "FETCH_SIZE"), (0, 0))[0]
    W = w.get((k, "WRITE_SIZE"), (0, 0))[0]
    lines.append(f"{k},{F:.1f},{W:.1f},{(2 * F + W) * 1024:.0f},{f[(k, 'FETCH_SIZE')][1]}")
lines += ["", "kernel,counter,avg_per_launch"]
for (k, c), (v, n) in sorted(q.items()):
    if k.startswith("cg1::k_accumulate") or k.startswith("cg1::k_seg") or k.startswith("cg1::k_bit"):
        lines.append(f"{k},{c},{v:.0f}")
open(out + "_pmc_summary.csv", "w").write("\n".join(lines) + "\n")
F, W = f[("cg1::k_accumulate", "FETCH_SIZE")][0], w[("cg1::k_accumulate", "WRITE_SIZE")][0]
bench = json.loads([ln for ln in open(f"gpurun_out/{tag}_bench.json") if ln.startswith("{")][0])
madds = bench["roofline_int_mad"]["mixed_adds_per_launch"]
valu, waves = q[("cg1::k_accumulate", "SQ_INSTS_VALU")][0], q[("cg1::k_accumulate", "SQ_WAVES")][0]
shutil.copy(glob.glob(f"gpurun_out/{tag}_vkt/*/*_kernel_stats.csv")[0], out + "_verify_kernel_stats.csv")
shutil.copy(f"gpurun_out/{tag}_sweep.txt", out + "_sweep.txt")
shutil.copy(f"gpurun_out/{tag}_vkt.txt", out + "_bench_verify.txt")
json.dump({"source": out + "_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over python3 bench.py --steps 4 --warmup 1)",
           "sq_source": out + "_pmc_summary.csv (rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ..., its own pass); mixed adds counted on the device in "
                        + out + "_bench.json; a wave-level mixed add = 64 lane-level ones (chunks are length-ordered: lanes of a wave finish together)",
           "workload": "single MSM of 2^20 terms, c=16, 1 GPU",
           "k_accumulate_SQ_INSTS_VALU": valu, "k_accumulate_SQ_WAVES": waves, "k_accumulate_mixed_adds": madds,
           "k_accumulate_valu_per_wave_mixed_add": valu / (madds / 64.0),
           "k_accumulate_FETCH_SIZE_KiB": F, "k_accumulate_WRITE_SIZE_KiB": W,
           "k_accumulate_hbm_bytes_per_launch": (2 * F + W) * 1024,
           "note": "FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; these fabric-side counters include "
                   "Infinity-Cache hits (the 128 MiB prepared-point array fits the 256 MiB cache), so true HBM traffic is lower. "
                   "Expected from the access pattern: 16 windows x 2^20 gathers x 128 B = 2.15 GB."},
          open("profiles/pmc_traffic.json", "w"), indent=1)
print(open(out + "_kernel_stats.csv").read()[:1500])
print(json.load(open("profiles/pmc_traffic.json"))["k_accumulate_hbm_bytes_per_launch"])
