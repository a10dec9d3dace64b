"""experiment: two half batches (25 000 reads each, full 64-lane waves) run from two host threads on their own streams, started together
or a fraction of a launch apart, against one batch of 50 000 -- does work of different phases in flight at the same time relieve the
fill's write path?  (GA_LANES_SPREAD=0: 391 + 391 waves fit next to each other.)  Prints ms per 50 000 reads of each arrangement.
Measured (round 3, after GA_LANES_SPREAD was wired up again -- an earlier run without it had both halves spread over all wave slots
and told nothing): one batch of 50 000 in 782 full waves 29.4 ms; two halves next to each other 32.1-33.7 ms per 50 000 reads whatever
the offset (0 / 8 / 15 / 22 ms).  Work of different phases in flight at once did not pay here: each half runs the same instruction
stream for half the reads per SIMD-slot pair, which costs more than the relieved write path gives back."""
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
