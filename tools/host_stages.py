#!/usr/bin/env python3
"""Host-side stages of one batch (bench.py workload): job building + upload, kernels, download + assembly."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphaligner_amd import binding, synth

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    g = synth.linear_graph(4641652, node_len=64, seed=42)
    reads, seeds = synth.simulate_reads(g, n, 10000, sub=0.04, ins=0.04, dele=0.04, seed=43)
    graph = binding.Graph(gfa=g.gfa())
    bp = sum(len(r) for r in reads)
    best = None
    rs = binding.ReadSet(reads, seeds)          # the arrays the C ABI takes (built once: a C / C++ caller holds them already)
    for rep in range(4):
        t0 = time.perf_counter(); b = graph.prepare(rs, None, 35)
        t1 = time.perf_counter(); b.run()
        t2 = time.perf_counter(); res = b.collect(summary=True)
        t3 = time.perf_counter()
        cur = dict(prepare_s=round(t1 - t0, 3), run_s=round(t2 - t1, 3), collect_s=round(t3 - t2, 3), total_s=round(t3 - t0, 3), Gbp_s=round(bp / (t3 - t0) / 1e9, 3), aligned=int((res["failed"] == 0).sum()))
        if best is None or cur["total_s"] < best["total_s"]: best = cur
        del b
    print(json.dumps(best))

if __name__ == "__main__":
    main()
