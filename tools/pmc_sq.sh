set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_sq
rm -rf $O && mkdir -p $O
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/a -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/a.json 2> $O/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $O/b -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/b.json 2> $O/b.err
python3 - <<PY
import csv, glob
for d in ('a','b'):
    rows = list(csv.DictReader(open(glob.glob('$O/%s/*/*_counter_collection.csv' % d)[0])))
    tot = {}
    for r in rows:
        if 'ga_extend_kernel' in r['Kernel_Name']:
            tot[r['Counter_Name']] = tot.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for k, v in sorted(tot.items()): print(k, v)
PY
