# round-end evidence: kernel trace stats, HBM traffic counters (separate passes), and the bench lines
# usage (on the GPU box): bash tools/profile_round.sh
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/stats_bench.json 2> $O/stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/fetch_bench.json 2> $O/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/write_bench.json 2> $O/write.err
echo "write done"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
python3 bench.py --graph bubbles --genome 12100000 --reads 20000 --cpu-sample 0 --steps 2 --warmup 1 > $O/bench_bubbles.json 2> $O/bench_bubbles.err
python3 bench.py --graph dense --node-len 32 --genome 3000000 --reads 16000 --read-len 15000 --errors 0.02,0.08,0.05 --cpu-sample 0 --steps 2 --warmup 1 > $O/bench_dense.json 2> $O/bench_dense.err
find $O -name "*.csv" | head -20
