# counters of the lanes = reads kernel on the benchmark batch: instruction mix, busy / wait cycles, HBM traffic, L2 hits
# (each --pmc set is its own run; no tracing domains next to counters).  usage on the GPU box: bash tools/pmc_lanes.sh [bench args]
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_lanes
rm -rf $O && mkdir -p $O
run() { name=$1; shift; pmc=$1; shift
  rocprofv3 --pmc $pmc --output-format csv -d $O/$name -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --accuracy 0 --pipeline-chunks 0 --kernel-only "$@" > $O/$name.json 2> $O/$name.err; echo "$name done"; }
run a "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "$@"
run b "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES" "$@"
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE" "$@"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@"
run grbm "GRBM_GUI_ACTIVE" "$@"
python3 - <<PY
import csv, glob, json
out = {}
for d in ('a', 'b', 'fetch', 'write', 'tcc', 'grbm'):
    fs = glob.glob('$O/%s/*/*_counter_collection.csv' % d)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        if 'ga_lanes_kernel' in r['Kernel_Name']:
            out[r['Counter_Name']] = out.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
b = json.loads(open('$O/a.json').read().strip().splitlines()[-1])
cols = b['roofline']['column_updates_per_launch']
out['column_updates'] = cols
out['kernel_ms_under_counters'] = b['roofline']['kernel_ms']
if 'FETCH_SIZE' in out: out['hbm_read_bytes_uncorrected'] = out['FETCH_SIZE'] * 1024
if 'WRITE_SIZE' in out: out['hbm_write_bytes'] = out['WRITE_SIZE'] * 1024
out['per_column_update'] = {k: round(v / cols, 3) for k, v in out.items() if k.startswith('SQ_') or k.startswith('TCC')}
json.dump(out, open('$O/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
# the HBM traffic per column update that bench.py scales to its own launch (-> profiles/r3_hbm_traffic.json)
if 'FETCH_SIZE' in out and 'WRITE_SIZE' in out:
    rd, wr = out['FETCH_SIZE'] * 1024, out['WRITE_SIZE'] * 1024
    t = {
     "command": "bash tools/pmc_lanes.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of: python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0)",
     "kernel": b['roofline']['kernel'],
     "build": "$(cat build/build_id.txt 2>/dev/null || git rev-parse --short HEAD 2>/dev/null || echo unknown)",
     "column_updates_per_launch": cols,
     "FETCH_SIZE_KB": out['FETCH_SIZE'], "WRITE_SIZE_KB": out['WRITE_SIZE'],
     "correction": "gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM); this kernel's reads are 4-24 B per lane, which that guide calls uncalibrated; the x2 figure is used as the conservative one",
     "hbm_read_bytes_uncorrected": rd, "hbm_read_bytes_x2": 2 * rd, "hbm_write_bytes": wr,
     "hbm_bytes_per_column_update": (2 * rd + wr) / cols,
     "hbm_bytes_per_column_update_uncorrected": (rd + wr) / cols,
     "algorithmic_bytes_per_column_update": 28.0,
     "kernel_ms_under_counters": b['roofline']['kernel_ms'],
     "note": "writes: the arena takes 24 B per band column in blocks of 8 columns per lane and node (only the blocks of lanes that have columns there are written), 4 B end words, per-slice headers and node lists, the traceback's node runs; reads: the traceback's column windows (24 B records in 192 B blocks), previous end words, graph records"
    }
    json.dump(t, open('$O/hbm_traffic.json', 'w'), indent=1)
PY
