# A/B of the epoch ring (SMRT_HIP_EPOCH_RING) on short-epoch cases and the bench: bash tools/ring_ab.sh (on the GPU box)
for rep in 1 2; do
for v in 1 0; do
  echo "== SMRT_HIP_EPOCH_RING=$v" >> gpurun_out/ring.log
  SMRT_HIP_EPOCH_RING=$v timeout -k 10 200 python -u tools/bench_configs.py --cases train_n1_b256,train_n2_b128,balance_c1,balance_c2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'batch' in r: print(r['case'], round(r['ms_per_epoch_wall']*1e3, 1), 'us/epoch', round(r['samples_per_s']/1e6, 3), 'M/s')
    else: print(r['case'], {k: v for k, v in r.items() if 'wall' in k or k.endswith('_s')})
" >> gpurun_out/ring.log 2>&1 || exit 1
  SMRT_HIP_EPOCH_RING=$v timeout -k 10 200 python -u bench.py 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms_per_step', r['ms_per_step'], 'value', r['value'])
" >> gpurun_out/ring.log 2>&1 || exit 1
done; done
