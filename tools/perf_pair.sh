# linear headline + dense (chr22-like) in one go; prints value / kernel ms / detail
set -e
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/pp_linear.json 2> gpurun_out/pp_linear.err
timeout -k 10 400 python bench.py --graph dense --node-len 32 --genome 3000000 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05 --cpu-sample 0 --steps 2 --warmup 1 > gpurun_out/pp_dense.json 2> gpurun_out/pp_dense.err
python - <<PY
import json
for n in ('linear','dense'):
    d=json.loads(open('gpurun_out/pp_%s.json' % n).read().strip().splitlines()[-1])
    print(n, d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['detail']['jobs_retried_wide'], d['detail']['reads_failed'])
PY
