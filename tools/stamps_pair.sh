set -e
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --stamps > gpurun_out/st_linear.json 2> gpurun_out/st_linear.err
timeout -k 10 400 python bench.py --graph dense --node-len 32 --genome 3000000 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05 --cpu-sample 0 --steps 2 --warmup 1 --stamps > gpurun_out/st_dense.json 2> gpurun_out/st_dense.err
python - <<PY
import json
for n in ('linear','dense'):
    d=json.loads(open('gpurun_out/st_%s.json' % n).read().strip().splitlines()[-1])
    print(n, d['value'], d['roofline']['kernel_ms'], d['detail']['phase_share'], d['detail']['cycles_per_job'])
PY
