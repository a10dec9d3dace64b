set -e
for N in 1 0; do
  GA_NARROW=$N timeout -k 10 400 python bench.py --graph dense --node-len 32 --genome 3000000 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05 --cpu-sample 0 --steps 2 --warmup 1 > gpurun_out/dense_narrow$N.json 2> gpurun_out/dense_narrow$N.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/dense_narrow$N.json').read().strip().splitlines()[-1])
print('narrow=$N', d['value'], d['roofline']['kernel_ms'], d['detail'])
PY
done
