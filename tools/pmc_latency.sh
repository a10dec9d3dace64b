# in-flight levels of the lanes = reads kernel: SQ_INST_LEVEL_x / SQ_INSTS_x = mean latency of a memory / LDS instruction in cycles;
# LDS bank conflicts.  usage on the GPU box: bash tools/pmc_latency.sh [bench args]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_lat
rm -rf $O && mkdir -p $O
run() { name=$1; shift; pmc=$1; shift
  rocprofv3 --pmc $pmc --output-format csv -d $O/$name -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --accuracy 0 --pipeline-chunks 0 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run a "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM" "$@"
run b "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "$@"
run c "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "$@"
run d "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_IFETCH_LEVEL" "$@"
python3 - <<PY
import csv, glob, json
out = {}
for d in 'abcd':
    fs = glob.glob('$O/%s/*/*_counter_collection.csv' % d)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        if 'ga_lanes_kernel' in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
json.dump(out, open('$O/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
