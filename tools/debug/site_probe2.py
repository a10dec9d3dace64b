"""diagnostic: 8-bp node chain, 50 kb reads, lanes kernel first (library built with -DGA_DEBUG_SITE)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from graphaligner_amd import binding, synth
lib = sys.argv[1] if len(sys.argv) > 1 else None
graph = synth.linear_graph(400000, node_len=8, seed=5)
reads, seeds = synth.simulate_reads(graph, 6, 50000, seed=6)
g = binding.Graph(graph.nodes, graph.edges, lib_path=lib)
for flags in (1, 0):
    b = g.prepare(reads, seeds, 35, 0, flags)
    b.run()
    res = b.collect()
    st = b.stats()
    print("flags", flags, "status", [r["status"] for r in res], "pass", [r.get("kernel_pass") for r in res], "score", [r["score"] for r in res], "stamps", st["stamps"])
