#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ instruction-mix counters of the bench kernels (own pass, <= 8 SQ counters).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_sq
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-balance --no-diag"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o bench -- python3 $ARGS > $OUT/sq.log 2>&1
python3 - <<'PY'
import csv, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open("gpurun_out/prof_sq/sq/bench_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("smrt::", "")
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); n[k] += 1
for k in tot:
    if "fused" in k:
        print(k, "launches", n[k], {c: round(v / n[k], 1) for c, v in tot[k].items()})
PY
