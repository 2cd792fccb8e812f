#!/bin/bash
# Collect the round's evidence on the GPU box:  gpurun -- 'bash tools/collect_profiles.sh TAG'
# Writes gpurun_out/TAG_*; tools/summarize_profiles.py (run afterwards, anywhere) copies summaries into profiles/.
# Counters are collected in their own passes (no --kernel-trace / --stats together with --pmc).
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 400 python $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 300 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-python-face > $O/${TAG}_kt.txt 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-python-face > $O/${TAG}_fetch.txt 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-python-face > $O/${TAG}_write.txt 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_sq -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-python-face > $O/${TAG}_sq.txt 2>&1
# the metric's second half: kernel trace of the verify stream (distinct-proof fixture, batches of 1024)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_vkt -- python3 $R/bench.py --mode verify --steps 20 --warmup 3 --no-cpu-baseline > $O/${TAG}_vkt.txt 2>&1
# size sweep (per-phase table for the mid-size configs)
timeout -k 10 200 python $R/tools/gpu_sweep.py > $O/${TAG}_sweep.txt 2>&1
ls $O | grep $TAG
