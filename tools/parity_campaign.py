#!/usr/bin/env python3
"""Randomised parity campaign: many small graphs x reads through the C ABI (GPU library by default,
--emul for the host emulation of the device program) and through the CPU oracle; every field of
every result is compared (tests/parity_common.compare_read).  Prints a JSON summary."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(argv=None, quiet=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=60)
    ap.add_argument("--reads", type=int, default=24)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--emul", action="store_true")
    ap.add_argument("--no-trace", action="store_true", help="flags = 0: no TraceItem lists (the result assembly's forward-only path for seeds at the first base)")
    ap.add_argument("--only", type=int, nargs="*", default=None, help="align only these trials (the others are still generated, so the random stream is the same)")
    args = ap.parse_args(argv)
    import numpy as np
    from graphaligner_amd import synth
    import parity_common as pc
    lib = pc.emul_lib_path() if args.emul else None
    rng = np.random.default_rng(args.seed)
    stats = dict(reads=0, compared=0, mismatches=0, cyclic_trials=0, ramp_trials=0, fan_trials=0, sparse_slices=0, override_traces=0, dev_status={}, first_mismatches=[])
    t0 = time.time()
    for trial in range(args.trials):
        nl = int(rng.choice([3, 8, 16, 32, 64, 100]))
        snp = int(rng.choice([0, 25, 60, 100]))
        indel = int(rng.choice([0, 100, 400, 1000]))
        sv = int(rng.choice([0, 0, 2500]))
        L = int(rng.choice([300, 700, 1500, 3000, 6000]))
        bw = int(rng.choice([2, 8, 20, 35, 35, 35, 50, 90]))
        err = float(rng.choice([0.0, 0.01, 0.04, 0.04, 0.08]))
        mid = bool(rng.random() < 0.4)
        cyclic = trial % 3 == 1          # tandem-repeat back edges and self loops: bands with strongly connected components
        ramp = 0
        if trial % 4 == 2:               # -B: narrow band first, ramp width on demand (damaged reads make the HMM flip)
            bw = int(rng.choice([3, 5, 10, 15]))
            ramp = bw + int(rng.choice([15, 30, 60]))
        fan = trial % 7 == 5             # a stem ending in many long branches: bands of >= 200 000 cells (sparse method, backtrace override)
        try:
            if fan:
                branches = int(rng.choice([5, 8, 12, 40]))
                blen = int(rng.choice([60000, 30000, 20000, 6000])) if branches < 40 else 6000
                if branches * blen < 230000:
                    blen = 230000 // branches + 1000
                fg = synth.FanGraph(head_len=int(rng.choice([100, 200, 400])), stem_len=int(rng.choice([300, 600, 1200])), n_branches=branches, branch_len=blen,
                                    shared=int(rng.choice([0, 100, 300, 600])), seed=trial)
                g = fg
                reads, seeds = [], []
                for k in range(min(args.reads, 8)):
                    if k % 3 != 2:
                        r, sd = fg.read_through(int(rng.integers(0, branches)), 0, int(rng.choice([1200, 2000, 3000])), rng, sub=err, ins=err, dele=err)
                    else:
                        b2 = int(rng.integers(0, branches))
                        depth = int(rng.integers(300, 1500))
                        path = np.concatenate([fg.head, fg.stem, fg.branches[b2][:depth + 800]])
                        cut = len(fg.head) + len(fg.stem) + depth
                        pre = synth.add_errors(path[:cut], err, err, err, rng).tobytes().decode()
                        r, sd = pre + synth.add_errors(path[cut:], err, err, err, rng).tobytes().decode(), (3 + b2, len(pre), False)
                    reads.append(r); seeds.append(sd)
                if ramp == 0 and trial % 2 == 1:
                    ramp = bw + 25
            elif cyclic:
                g = synth.cyclic_graph(int(rng.choice([4000, 9000])), node_len=max(nl, 4), seed=trial, back_edges=int(rng.integers(2, 12)), self_loops=int(rng.integers(0, 4)),
                                       max_span=int(rng.integers(1, 9)), snp_every=snp if snp else 60)
                reads, seeds = synth.walk_reads(g, args.reads, min(L, 3000), sub=err, ins=err, dele=err, seed=trial, mid_seed=mid, first_nodes=max(1, len(g.nodes) // 3))
            else:
                g = synth.SynthGraph(synth.random_genome(int(rng.choice([6000, 15000, 40000])), 7000 + trial), node_len=nl, snp_every=snp, indel_every=indel, sv_every=sv, seed=trial)
                reads, seeds = synth.simulate_reads(g, args.reads, L, sub=err, ins=err, dele=err, seed=trial, mid_seed=mid)
        except RuntimeError:
            continue
        stats["fan_trials"] += int(fan)
        stats["cyclic_trials"] += int(cyclic and not fan)
        stats["ramp_trials"] += int(ramp > 0)
        if ramp:
            import parity_cases
            reads = parity_cases.damaged_reads(reads, rng)
        # sprinkle IUPAC / N / lower case into some reads
        for k in range(0, len(reads), 5):
            b = bytearray(reads[k].encode())
            for _ in range(8):
                b[int(rng.integers(len(b)))] = ord("NRYKMSWBDVnacgt"[int(rng.integers(15))])
            reads[k] = b.decode()
        if args.only is not None and trial not in args.only:
            continue
        if trial % 25 == 0:
            print("trial %d of %d, %d reads, %d mismatches, %.0f s" % (trial, args.trials, stats["reads"], stats["mismatches"], time.time() - t0), file=sys.stderr, flush=True)
        devs, oras = pc.run_both(g.nodes, g.edges, reads, seeds, bw, ramp=ramp, lib_path=lib, trace=not args.no_trace)
        if args.no_trace:
            oras = [dict(o, trace=np.zeros((0, 7), dtype=np.int64)) for o in oras]
        for i, (d, o) in enumerate(zip(devs, oras)):
            stats["reads"] += 1
            stats["sparse_slices"] += o.get("sparse_slices", 0)
            stats["override_traces"] += o.get("override_traces", 0)
            stats["dev_status"][str(d["status"])] = stats["dev_status"].get(str(d["status"]), 0) + 1
            if d["status"] in (10, 11, 12, 13, 14):
                continue      # band beyond the widest kernel variant: reported as a capacity status, never a silently different answer
            stats["compared"] += 1
            try:
                pc.compare_read(d, o, "trial %d read %d" % (trial, i))
            except AssertionError as e:
                stats["mismatches"] += 1
                if len(stats["first_mismatches"]) < 5:
                    stats["first_mismatches"].append(dict(trial=trial, node_len=nl, snp=snp, indel=indel, sv=sv, L=L, bw=bw, err=err, mid=mid, what=str(e)[:300]))
    stats["seconds"] = round(time.time() - t0, 1)
    if quiet:
        return stats
    print(json.dumps(stats))
    return 1 if stats["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
