#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ counters of the persistent epoch kernel on the bench net (tools/stamp_persist.py, 128
# mini-batches per launch), two passes of <= 8 SQ counters.  $1 = tag; SMRT_HIP_LIB selects the build under test.
set -e
TAG=${1:-head}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_persist_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -o p -- python3 tools/stamp_persist.py 16384 > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM --output-format csv -d $OUT/b -o p -- python3 tools/stamp_persist.py 16384 > $OUT/b.log 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, collections, json, sys, glob
out, tag = sys.argv[1], sys.argv[2]
res = {"command": "tools/pmc_persist.sh %s (rocprofv3 --pmc, two passes; python3 tools/stamp_persist.py 16384: 128 mini-batches of 128 rows per launch)" % tag,
       "units": "per launch of persist_train_kernel, summed over its waves (8 compute + 28 owner workgroups of 8 waves)"}
for p in "ab":
    tot = collections.defaultdict(float); seen = set()
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "persist_train" not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
    for c, v in tot.items(): res[c] = round(v / max(1, len(seen)), 1)
    res["launches_" + p] = len(seen)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
