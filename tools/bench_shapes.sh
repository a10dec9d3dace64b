#!/bin/bash
# the three workload shapes of profiles/*_bench_lines.json on one box: linear (the headline), yeast-like bubbles, chr22-like dense; the last
# two once with the library's own first-pass choice and once with the lanes = reads kernel forced.  Output: gpurun_out/shapes_<tag>.txt
tag=${1:-run}
mkdir -p gpurun_out
line() {
  name=$1; shift
  timeout -k 10 600 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --accuracy 0 --check 32 --pipeline-chunks 0 "$@" > gpurun_out/shape_${tag}_$name.json 2> gpurun_out/shape_${tag}_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/shape_${tag}_$name.json').read().strip().splitlines()[-1])
    print('$name', 'value', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'all_passes_ms', d['detail']['all_passes_ms'], 'ladder', d['detail']['jobs_left_to_the_wave_per_read_ladder'], 'failed', d['detail']['reads_failed'], 'kernel_only', d['detail']['kernel_only_Gbp_s'])
except Exception as e:
    print('$name', 'FAILED', e)
PY
}
line linear
line bubbles --graph bubbles --genome 12100000 --reads 20000
GA_LANES=1 line bubbles_lanes --graph bubbles --genome 12100000 --reads 20000
line dense --graph dense --genome 3000000 --node-len 32 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05
GA_LANES=1 line dense_lanes --graph dense --genome 3000000 --node-len 32 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05
