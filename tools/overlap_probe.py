#!/usr/bin/env python3
"""Do a batch's kernels and another batch's result assembly really overlap?  Times run alone, collect alone, and the two at once."""
import json, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphaligner_amd import binding, synth

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    if len(sys.argv) > 2 and sys.argv[2] == "torch-first":       # (as bench.py does: torch before the library touches the device)
        import torch
        print("torch threads", torch.get_num_threads(), "cuda", torch.cuda.is_available())
        torch.cuda.set_device(0)
    g = synth.linear_graph(4641652, node_len=64, seed=42)
    reads, seeds = synth.simulate_reads(g, n, 10000, sub=0.04, ins=0.04, dele=0.04, seed=43)
    graph = binding.Graph(gfa=g.gfa())
    rs = binding.ReadSet(reads, seeds)
    a = graph.prepare(rs, None, 35)
    b = graph.prepare(rs, None, 35)
    for x in (a, b):
        x.run(); x.collect(summary=True)
    out = {}
    t0 = time.perf_counter(); a.run(); out["run_alone_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    t0 = time.perf_counter(); a.collect(summary=True); out["collect_alone_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    b.run()
    # a's kernels and b's assembly at once
    times = {}
    def do_collect():
        t = time.perf_counter(); b.collect(summary=True); times["collect"] = (time.perf_counter() - t) * 1e3
    th = threading.Thread(target=do_collect)
    t0 = time.perf_counter()
    th.start()
    t = time.perf_counter(); a.run(); times["run"] = (time.perf_counter() - t) * 1e3
    th.join()
    out["both_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    out["run_while_collect_ms"] = round(times["run"], 1); out["collect_while_run_ms"] = round(times["collect"], 1)
    out["kernel_ms_of_that_run"] = round(a.stats()["kernel_ms"], 1)
    print(json.dumps(out))
    # the bench's loop: two resident batches take turns, the previous step's results are assembled while this step's kernels run
    from concurrent.futures import ThreadPoolExecutor
    def steps(n, log):
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = None
            for k in range(n):
                x = a if k % 2 == 0 else b
                t = time.perf_counter(); x.run(); log.append(("run", k, t, time.perf_counter()))
                if pending is not None:
                    t = time.perf_counter(); pending.result(); log.append(("wait", k, t, time.perf_counter()))
                def job(x=x, k=k):
                    t = time.perf_counter(); r = x.collect(summary=True); log.append(("collect", k, t, time.perf_counter())); return r
                pending = pool.submit(job)
            pending.result()
    for with_torch in (False,):
        log = []
        steps(2, [])
        t0 = time.perf_counter(); steps(6, log); total = time.perf_counter() - t0
        print("torch imported:" if with_torch else "plain:", "6 steps in %.1f ms" % (total * 1e3))
        for what, k, ta, tb in sorted(log, key=lambda e: e[2]):
            print("   %-8s step %d  %7.1f .. %7.1f ms" % (what, k, (ta - t0) * 1e3, (tb - t0) * 1e3))

if __name__ == "__main__":
    main()
