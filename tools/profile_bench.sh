#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 passes over the default bench.py command.
#   1. --kernel-trace --stats          -> per-kernel durations
#   2. --pmc FETCH_SIZE                -> HBM read traffic   (own pass: FETCH_SIZE takes 3 of the 4 TCC slots)
#   3. --pmc WRITE_SIZE                -> HBM write traffic  (own pass)
# Outputs land under gpurun_out/prof_bench/; tools/pmc_traffic.py turns them into profiles/*.json|csv.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_bench
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-balance --no-diag"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python3 $ARGS > $OUT/write.log 2>&1
ls -R $OUT | head -30
