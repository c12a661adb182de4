#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ counters of fused_infer_kernel over tools/infer_only.py (two passes, <= 8 SQ counters each).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_infer_sq
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -o infer -- python3 tools/infer_only.py > $OUT/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -o infer -- python3 tools/infer_only.py > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, collections, json
out = {}
for p in ("a", "b"):
    tot = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open("gpurun_out/prof_infer_sq/%s/infer_counter_collection.csv" % p)):
        if "fused_infer" not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
    out.update({c: v / max(len(seen), 1) for c, v in tot.items()}); out["launches_" + p] = len(seen)
print(json.dumps(out))
json.dump(out, open("gpurun_out/prof_infer_sq/summary.json", "w"), indent=1)
PY
