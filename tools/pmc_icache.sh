cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_ic
rm -rf $O && mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_BRANCH[A-Z_]*\|SQ_INST_LEVEL_[A-Z]*\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_ANY" | sort -u | tr '\n' ' ' > $O/names.txt
cat $O/names.txt; echo
run() { name=$1; shift; pmc=$1; shift
  rocprofv3 --pmc $pmc --output-format csv -d $O/$name -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --accuracy 0 --pipeline-chunks 0 > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run a "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
run b "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
python3 - <<PY
import csv, glob, json
out = {}
for d in 'ab':
    fs = glob.glob('$O/%s/*/*_counter_collection.csv' % d)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        if 'ga_lanes_kernel' in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
print(json.dumps(out, indent=1))
PY
tail -3 $O/a.err
