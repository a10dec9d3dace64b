"""experiment: two half batches (25 000 reads each, full 64-lane waves) run from two host threads on their own streams, started together
or a fraction of a launch apart, against one batch of 50 000 -- does work of different phases in flight at the same time relieve the
fill's write path?  (GA_LANES_SPREAD=0: 391 + 391 waves fit next to each other.)  Prints ms per 50 000 reads of each arrangement.
Measured (round 3): one batch 32.1 ms; two halves 52.5-52.8 ms whatever the offset -- the two launches did not overlap at all (26 ms
each, one after the other), so the question is still open; why launches from two non-blocking streams of one process serialise here
is the first thing to find out."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["GA_LANES_SPREAD"] = "0"
from graphaligner_amd import binding, synth

g = synth.linear_graph(4641652, node_len=64, seed=42)
reads, seeds = synth.simulate_reads(g, 50000, 10000, sub=0.04, ins=0.04, dele=0.04, seed=43)
graph = binding.Graph(g.nodes, g.edges)
n = len(reads)
full = graph.prepare(reads, seeds, 35, 0, 0)
h1 = graph.prepare(reads[:n // 2], seeds[:n // 2], 35, 0, 0)
h2 = graph.prepare(reads[n // 2:], seeds[n // 2:], 35, 0, 0)
for b in (full, h1, h2):
    b.run()
K = int(os.environ.get("K", "6"))
t0 = time.perf_counter()
for _ in range(K):
    full.run()
print("one batch of 50 000 (64 reads per wave, 782 waves): %.2f ms per 50 000 reads" % ((time.perf_counter() - t0) / K * 1e3), flush=True)
for offset in (0.0, 0.008, 0.015, 0.022):
    def loop(b, delay):
        time.sleep(delay)
        for _ in range(K):
            b.run()
    ta = threading.Thread(target=loop, args=(h1, 0.0)); tb = threading.Thread(target=loop, args=(h2, offset))
    t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join()
    print("two halves of 25 000, the second %.0f ms later: %.2f ms per 50 000 reads (the offset included once in %d rounds)" % (offset * 1e3, (time.perf_counter() - t0) / K * 1e3, K), flush=True)
