#!/usr/bin/env python3
"""Long damaged reads with -B on the GPU against the oracle (ramp redo, stale checkpoints, arena growth in the general variants)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from graphaligner_amd import synth, binding
import parity_common as pc, parity_cases as cases

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    g = synth.bubble_graph(300000, node_len=32, seed=91)
    out = dict(reads=0, mismatches=0, status={}, first=[])
    for length, bw, ramp, seed in ((10000, 10, 60, 1), (6000, 5, 35, 2), (15000, 15, 80, 3)):
        reads, seeds = synth.simulate_reads(g, n // 3, length, seed=seed, mid_seed=(seed % 2 == 0))
        reads = cases.damaged_reads(reads, np.random.default_rng(seed))
        t0 = time.time()
        devs, oras = pc.run_both(g.nodes, g.edges, reads, seeds, bw, ramp=ramp)
        for i, (d, o) in enumerate(zip(devs, oras)):
            out["reads"] += 1
            k = "%d/%s" % (d["status"], o["message"][:40])
            out["status"][k] = out["status"].get(k, 0) + 1
            try:
                pc.compare_read(d, o, "len %d read %d" % (length, i))
            except AssertionError as e:
                out["mismatches"] += 1
                if len(out["first"]) < 5: out["first"].append(str(e)[:300])
    print(json.dumps(out))
    return 1 if out["mismatches"] else 0

if __name__ == "__main__":
    sys.exit(main())
