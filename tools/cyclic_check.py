#!/usr/bin/env python3
"""Long reads walking round tandem repeats / self loops on the GPU against the oracle (bands with strongly connected components)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from graphaligner_amd import synth, binding
import parity_common as pc

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out = dict(reads=0, mismatches=0, status={}, jobs_retried=0, first=[])
    for k, (nl, bw, back, selfl, span, length) in enumerate(((16, 35, 60, 10, 6, 10000), (32, 35, 40, 5, 3, 8000), (8, 20, 80, 20, 8, 5000))):
        g = synth.cyclic_graph(120000, node_len=nl, seed=700 + k, back_edges=back, self_loops=selfl, max_span=span, snp_every=80)
        reads, seeds = synth.walk_reads(g, n // 3, length, seed=k, mid_seed=(k == 1), first_nodes=max(1, len(g.nodes) // 3))
        devs, oras = pc.run_both(g.nodes, g.edges, reads, seeds, bw)
        gg = binding.Graph(g.nodes, g.edges)
        b = gg.prepare(reads, seeds, bw); b.run(); out["jobs_retried"] += b.stats()["jobs_retried"]
        for i, (d, o) in enumerate(zip(devs, oras)):
            out["reads"] += 1
            key = "%d/%s" % (d["status"], o["message"][:40])
            out["status"][key] = out["status"].get(key, 0) + 1
            try:
                pc.compare_read(d, o, "cfg %d read %d" % (k, i))
            except AssertionError as e:
                out["mismatches"] += 1
                if len(out["first"]) < 5: out["first"].append(str(e)[:300])
    print(json.dumps(out))
    return 1 if out["mismatches"] else 0

if __name__ == "__main__":
    sys.exit(main())
