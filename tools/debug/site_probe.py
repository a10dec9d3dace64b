"""diagnostic: which traceback assertion fires on the device (library built with -DGA_DEBUG_SITE)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from graphaligner_amd import binding, synth
lib = sys.argv[1] if len(sys.argv) > 1 else None
graph = synth.linear_graph(20000, node_len=64, seed=73)
reads, seeds = synth.simulate_reads(graph, 12, 1800, seed=9)
reads, seeds = reads[:1], seeds[:1]
g = binding.Graph(graph.nodes, graph.edges, lib_path=lib)
for flags in (0, 1):
    b = g.prepare(reads, seeds, 35, 0, flags)
    b.run()
    res = b.collect()
    st = b.stats()
    print("flags", flags, "status", [r["status"] for r in res], "stamps", st["stamps"], "variant", st.get("main_variant"))
