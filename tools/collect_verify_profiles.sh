#!/bin/bash
# Evidence for the batch shuffle verifier:  gpurun -- 'bash tools/collect_verify_profiles.sh TAG'
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 300 python $R/bench.py --mode verify --steps 40 --warmup 3 > $O/${TAG}_bench_verify.json 2> $O/${TAG}_bench_verify.err
timeout -k 10 300 python $R/bench.py --mode verify --verify-mode independent --steps 40 --warmup 3 > $O/${TAG}_bench_verify_independent.json 2>> $O/${TAG}_bench_verify.err
timeout -k 10 300 python $R/tools/gpu_verify_timing.py > $O/${TAG}_verify.txt 2>&1
timeout -k 10 120 python $R/tools/host_microbench.py >> $O/${TAG}_verify.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_vkt -- python3 $R/bench.py --mode verify --steps 5 --warmup 1 > $O/${TAG}_vkt.txt 2>&1
ls $O | grep $TAG
