"""A C5-shaped line: a variation graph of >= 1 Gbp with SNP / indel bubbles and nodes <= 32 bp, built natively on the GPU
box (tests/native/ga_scalegen.cpp: ga_scalegen_bubbles), 20 000 x 50 kb ONT-error reads walked through random alleles, one batch.
Prints one JSON line: Gbp/s (device passes, and reads-in-host-memory -> results-in-host-memory), roofline fraction, resident waves,
scratch; a sample of reads (the first, the last -- past directed node 2^27 when the graph is big enough -- and a few between) is
compared with the CPU oracle on a window of the same graph written out again as GFA.

    python tools/bench_c5.py [--gbp 1.0] [--reads 20000] [--read-len 50000] [--check 6]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbp", type=float, default=1.0)
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--read-len", type=int, default=50000)
    ap.add_argument("--block", type=int, default=45)
    ap.add_argument("--node-len", type=int, default=32)
    ap.add_argument("--bandwidth", type=int, default=35)
    ap.add_argument("--check", type=int, default=6)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import __graft_entry__ as entry
    entry.build_product()
    from graphaligner_amd import binding, synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "native")])
    SG = C.CDLL(os.path.join(ROOT, "tests", "_build", "libga_scalegen.so"))
    SG.ga_scalegen_bubbles.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
    SG.ga_scalegen_bubbles_gfa.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_char_p, C.c_uint64]
    SG.ga_scalegen_bubbles_gfa.restype = C.c_uint64
    SG.ga_scalegen_bubbles_walk.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_char_p]
    SG.ga_scalegen_bubbles_walk.restype = C.c_uint64
    seed = 50
    n_blocks = int(args.gbp * 1e9) // args.block
    L = binding.load()
    t0 = time.time()
    gg = object.__new__(binding.Graph)
    gg.L = L
    gg.h = L.ga_graph_create()
    binding._check(L, SG.ga_scalegen_bubbles(gg.h, seed, n_blocks, args.block, args.node_len), "ga_scalegen_bubbles")
    t_build = time.time() - t0
    n_directed = int(L.ga_graph_node_count(gg.h))
    t0 = time.time()
    binding._check(L, L.ga_graph_upload(gg.h, 0), "ga_graph_upload")
    t_upload = time.time() - t0
    print("graph: %d blocks, %d directed nodes, %.2f Gbp both strands; built in %.0f s, uploaded in %.1f s" % (n_blocks, n_directed, L.ga_graph_bp(gg.h) / 1e9, t_build, t_upload),
          file=sys.stderr, flush=True)
    rng = np.random.default_rng(52)
    blocks_per_read = args.read_len // (args.block - 2) + 64
    starts = rng.integers(4, n_blocks - blocks_per_read - 4, size=args.reads)
    starts[0] = 4
    starts[-1] = n_blocks - blocks_per_read - 8            # the last read lies at the graph's far end: node indices past 2^27 at full size
    reads, seeds = [], []
    buf = C.create_string_buffer(args.read_len)
    t0 = time.time()
    for i, b0 in enumerate(starts):
        n = SG.ga_scalegen_bubbles_walk(seed, 1000 + i, int(b0), n_blocks, args.block, args.node_len, args.read_len, buf)
        body = np.frombuffer(buf.raw[:n], dtype=np.uint8)
        reads.append(synth.add_errors(body, 0.04, 0.04, 0.04, rng).tobytes().decode())
        seeds.append((int(b0) * 16 + 1, 0, False))
    t_reads = time.time() - t0
    total_bp = sum(len(r) for r in reads)
    print("reads: %d, %.3f Gbp, generated in %.0f s" % (len(reads), total_bp / 1e9, t_reads), file=sys.stderr, flush=True)
    rs = binding.ReadSet(reads, seeds)
    t0 = time.time()
    batch = gg.prepare(rs, None, args.bandwidth, 0)
    t_prep = time.time() - t0
    batch.run()                                            # warm-up (scratch pool, first touch)
    k_ms, step_s = [], []
    for _ in range(args.steps):
        t0 = time.time()
        batch.run()
        t1 = time.time()
        summary = batch.collect(summary=True)
        step_s.append(time.time() - t0)
        k_ms.append(batch.stats()["kernel_ms"])
    st = batch.stats()
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    aligned = int(lens[summary["failed"] == 0].sum())
    kms = float(np.mean(k_ms))
    out = {
        "workload": "C5 shape: %.2f Gbp variation graph (SNP every %d bp, 1-5 bp indels, nodes <= %d bp, %d directed nodes) + %d x %d bp ONT-error reads, band=%d"
                    % (args.gbp, args.block, args.node_len, n_directed, args.reads, args.read_len, args.bandwidth),
        "kernel_only_Gbp_s": round(aligned / (kms * 1e-3) / 1e9, 4), "all_passes_ms": round(kms, 2), "first_pass_ms": round(st["main_kernel_ms"], 2),
        "first_pass_variant": int(st["main_variant"]), "resident_waves": int(st["slots"]), "waves_per_cu": int(st["waves_per_cu"]), "scratch_GB": round(st["scratch_bytes"] / 1e9, 2),
        "results_in_host_memory_Gbp_s": round(aligned / float(np.mean(step_s)) / 1e9, 4), "prepare_s": round(t_prep, 2),
        "column_updates": int(st["column_updates"]), "roofline_frac": round(28.0 * st["column_updates"] / (kms * 1e-3) / 1e9 / 8000.0, 5),
        "reads_failed": int((summary["failed"] != 0).sum()), "jobs_left_to_the_ladder": int(st["jobs_retried"]),
        "graph_build_s": round(t_build), "graph_upload_s": round(t_upload, 1),
    }
    # ---- the oracle on windows of the same graph ----
    if args.check > 0:
        import oracle_binding as ob
        import parity_common as pc
        pick = sorted(set([0, len(reads) - 1] + [int(x) for x in np.linspace(1, len(reads) - 2, max(0, args.check - 2))]))
        res = gg.align([reads[i] for i in pick], [seeds[i] for i in pick], args.bandwidth, 0)
        for d, i in zip(res, pick):
            b0 = int(starts[i])
            lo, hi = max(0, b0 - 4), min(n_blocks, b0 + blocks_per_read + 64)
            n = SG.ga_scalegen_bubbles_gfa(seed, lo, hi, args.block, args.node_len, None, 0)
            text = C.create_string_buffer(int(n))
            SG.ga_scalegen_bubbles_gfa(seed, lo, hi, args.block, args.node_len, text, n)
            segs, links = [], []
            for line in text.raw.decode().split("\n"):
                f = line.split("\t")
                if f[0] == "S":
                    segs.append((int(f[1]), f[2]))
                elif f[0] == "L":
                    links.append((int(f[1]), False, int(f[3]), False))
            og = ob.OracleGraph.from_gfa_segments(segs, links, 0)
            o = og.align(reads[i], [seeds[i]], args.bandwidth)
            pc.compare_read(dict(d, trace=np.zeros((0, 7), dtype=np.int64)), dict(o, trace=np.zeros((0, 7), dtype=np.int64)), "C5 read %d" % i)
        out["oracle_checked_reads"] = len(pick)
        out["last_read_first_node_index"] = int(starts[-1]) * 16 * 2
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
