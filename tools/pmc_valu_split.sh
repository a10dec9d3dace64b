# vector-instruction count per launch with and without the traceback (diagnostic switch), linear and dense workloads
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_split
rm -rf $O && mkdir -p $O
run() { # name, extra env, bench args
  name=$1; shift
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/$name -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 "$@" > $O/$name.json 2> $O/$name.err
}
run lin_full
GA_DIAG_NO_TRACEBACK=1 run lin_notb
run dense_full --graph dense --node-len 32 --genome 3000000 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05
GA_DIAG_NO_TRACEBACK=1 run dense_notb --graph dense --node-len 32 --genome 3000000 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05
python3 - <<PY
import csv, glob, json
for n in ('lin_full','lin_notb','dense_full','dense_notb'):
    f = sorted(glob.glob('$O/%s/*/*_counter_collection.csv' % n))[-1]
    tot = {}
    for r in csv.DictReader(open(f)):
        if 'ga_extend_kernel<32' in r['Kernel_Name']:
            tot[r['Counter_Name']] = tot.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    d = json.loads(open('$O/%s.json' % n).read().strip().splitlines()[-1])
    cols = d['roofline']['column_updates_per_launch']
    print(n, 'columns', cols, 'kernel_ms', d['roofline']['kernel_ms'], {k: round(v / cols, 2) for k, v in sorted(tot.items())})
PY
