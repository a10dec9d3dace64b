#!/bin/bash
# same-box A/B of product builds: tools/ab_bench.sh name1 name2 ... (build/ab/libga_<name>.so, made by tools/build_variant.sh); extra bench.py
# flags through AB_FLAGS.  Prints kernel ms / Gbp/s per variant; every run also spot-checks reads against the oracle.
set -e
mkdir -p gpurun_out
for n in "$@"; do
  timeout -k 10 300 python bench.py --steps ${AB_STEPS:-3} --warmup 1 --cpu-sample 0 --accuracy 0 --check 16 --lib build/ab/libga_$n.so $AB_FLAGS > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_$n.json').read().strip().splitlines()[-1])
print('$n', 'Gbp/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'failed', d['detail']['reads_failed'], 'spot', d['detail'].get('oracle_spot_check_reads'))
PY
done
