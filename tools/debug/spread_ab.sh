for i in 1 2; do
for sp in 1 52 56 60 0; do
GA_LANES_SPREAD=$sp timeout -k 10 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --accuracy 0 --check 32 --pipeline-chunks 0 --kernel-only > gpurun_out/sp.json 2> gpurun_out/sp.err
python - <<PY
import json
d=json.loads(open('gpurun_out/sp.json').read().strip().splitlines()[-1])
print('spread=$sp kernel_ms', d['roofline']['kernel_ms'], 'waves', d['detail']['waves'], 'scratch', d['detail']['scratch_GB'], 'failed', d['detail']['reads_failed'], 'spot', d['detail'].get('oracle_spot_check_reads'))
PY
done; done
