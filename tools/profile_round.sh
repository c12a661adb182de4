#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the round's rocprofv3 evidence in one call, every pass a run of its own under its own time limit.
#   gpurun_out/r03/bench_{trace,fetch,write}   the default bench.py command (tools/profile_bench.sh's three passes)
#   gpurun_out/r03/pmc_persist_r03             SQ counters of the persistent kernel (tools/pmc_persist.sh)
#   gpurun_out/r03/n1_b128 | n1_b256 | c3      --kernel-trace --stats of N1 at the reference's batches and of the C3 call
#   gpurun_out/r03/n1_b8192 | n1_b65536       the same for N1 at large mini-batches (what a data-parallel C4 rank runs)
#   gpurun_out/r03/sq_n1_b128 | sq_n1_b65536 | sq_infer    SQ counters of the small-batch kernels, dw_kernel / gemm_kernel, fused_infer
# tools/collect_profiles.py turns them into profiles/r03_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r03
rm -rf $R && mkdir -p $R
bash tools/profile_bench.sh > $R/profile_bench.log 2>&1 && mv gpurun_out/prof_bench $R/bench
bash tools/pmc_persist.sh r03 > $R/pmc_persist.log 2>&1 && mv gpurun_out/pmc_persist_r03 $R/
T="timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv"
$T -d $R/n1_b128 -o p -- python3 tools/bench_configs.py --cases train_n1_b128 > $R/n1_b128.log 2>&1
$T -d $R/n1_b256 -o p -- python3 tools/bench_configs.py --cases train_n1_b256 > $R/n1_b256.log 2>&1
$T -d $R/c3 -o p -- python3 tools/c3_timing.py > $R/c3.log 2>&1
$T -d $R/n1_b8192 -o p -- python3 tools/n1_large.py 8192 > $R/n1_b8192.log 2>&1
$T -d $R/n1_b65536 -o p -- python3 tools/n1_large.py 65536 > $R/n1_b65536.log 2>&1
P="timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv"
Q="timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv"
$P -d $R/sq_n1_b128/a -o p -- python3 tools/bench_configs.py --cases train_n1_b128 > $R/sq_n1_b128_a.log 2>&1
$Q -d $R/sq_n1_b128/b -o p -- python3 tools/bench_configs.py --cases train_n1_b128 > $R/sq_n1_b128_b.log 2>&1
$P -d $R/sq_n1_b65536/a -o p -- python3 tools/n1_large.py 65536 > $R/sq_n1_b65536_a.log 2>&1
$Q -d $R/sq_n1_b65536/b -o p -- python3 tools/n1_large.py 65536 > $R/sq_n1_b65536_b.log 2>&1
$P -d $R/sq_infer/a -o p -- python3 tools/infer_only.py > $R/sq_infer_a.log 2>&1
$Q -d $R/sq_infer/b -o p -- python3 tools/infer_only.py > $R/sq_infer_b.log 2>&1
python3 bench.py > $R/bench_line.json 2> $R/bench_line.err
python3 tools/bench_configs.py > $R/configs.jsonl 2> $R/configs.err || true
du -sh $R; ls $R
