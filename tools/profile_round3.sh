# round-3 evidence: kernel trace stats of the default bench, counters of the lanes = reads kernel (separate --pmc passes), and the
# bench lines of the three graph shapes.  usage on the GPU box: bash tools/profile_round3.sh
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof3
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --check 0 --accuracy 0 --pipeline-chunks 0 --kernel-only > $O/stats_bench.json 2> $O/stats.err
echo "stats done"
bash tools/pmc_lanes.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc_lanes/summary.json $O/pmc_summary.json
cp gpurun_out/pmc_lanes/hbm_traffic.json $O/hbm_traffic.json
echo "pmc done"
cp $O/hbm_traffic.json profiles/r3_hbm_traffic.json      # (so that the lines below carry this build's traffic figure)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
python3 bench.py --graph bubbles --genome 12100000 --reads 20000 --cpu-sample 0 --check 32 --steps 6 --warmup 2 > $O/bench_bubbles.json 2> $O/bench_bubbles.err
GA_DEBUG_PASSES=1 python3 bench.py --graph dense --node-len 32 --genome 3000000 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05 --cpu-sample 0 --check 32 --steps 6 --warmup 2 > $O/bench_dense.json 2> $O/bench_dense.err
GA_LANES=1 python3 bench.py --graph bubbles --genome 12100000 --reads 20000 --cpu-sample 0 --check 32 --steps 6 --warmup 2 > $O/bench_bubbles_lanes_first.json 2> $O/bench_bubbles_lanes_first.err
GA_LANES=1 python3 bench.py --graph dense --node-len 32 --genome 3000000 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05 --cpu-sample 0 --check 32 --steps 6 --warmup 2 > $O/bench_dense_lanes_first.json 2> $O/bench_dense_lanes_first.err
python3 - <<PY
import json
lines = {}
for n in ('default', 'bubbles', 'dense', 'bubbles_lanes_first', 'dense_lanes_first'):
    try: lines[n] = json.loads(open('$O/bench_%s.json' % n).read().strip().splitlines()[-1])
    except Exception as e: lines[n] = {'error': str(e)}
json.dump(lines, open('$O/bench_lines.json', 'w'), indent=1)
for n, d in lines.items(): print(n, d.get('value'), d.get('roofline', {}).get('kernel_ms'), d.get('roofline', {}).get('frac'))
PY
find $O -name "*kernel_stats.csv" | head
