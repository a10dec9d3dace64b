#!/bin/bash
# stamped builds of the current tree (bench.py --stamps, GA_STAMPS_LEVEL = 1, 2, 4): shares of a wave's cycles per phase
cd "${GRAFT_REPO_ROOT:-.}"
for lv in 1 2 4; do
  GA_STAMPS_LEVEL=$lv timeout -k 10 400 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --check 0 --accuracy 0 --pipeline-chunks 0 --kernel-only --stamps > gpurun_out/st$lv.json 2> gpurun_out/st$lv.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/st$lv.json").read().strip().splitlines()[-1])
print("level $lv: kernel_ms", d["roofline"]["kernel_ms"]); print("  phase_share", json.dumps(d["detail"].get("phase_share"))); print("  stamps_raw", d["detail"].get("stamps_raw"))
PY
done
