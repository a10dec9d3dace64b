#!/usr/bin/env python3
"""bench.py -- aligned Gbp/s of the seed-and-extend hot path on MI355X.

One "step" = one pass of the extension kernel over one batch of synthetic reads that is
already resident in HBM (reads coded and uploaded, graph uploaded, before the timed region).

Workload at N=1 (BASELINE.json configs[1], restated synthetically as SURVEY.md 8(d) C2):
  E. coli-scale linear graph: 4,641,652 bp uniform ACGT (seed 42) as one chain of 64-bp nodes,
  50,000 reads x 10,000 bp drawn from it, SimulateReads-style errors s=i=d=0.04 (seed 43),
  both strands, one seed hit per read at read position 0, bandwidth 35, no ramp.
With --gpus N every rank runs its own 50,000-read shard (different read seed) against a graph
replicated in its GPU's HBM -- weak scaling, no collective on the data path.

Prints ONE JSON line (rank 0): metric/value per the driver contract plus `roofline`
(algorithmic 28 B per column update / live HIP-event time of the dominant kernel vs 8 TB/s HBM) and
`cpu_baseline` (the CPU oracle, multi-threaded, on a bounded sample of the same reads).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_COLUMN_UPDATE = 28.0      # 0.25 base + 4 prev end + 4 end out + 20 VP/VN/score
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_STREAM_GBS = 5318.6             # what a streaming copy reaches on this pool (profiles/r1_hbm_stream.txt, tools/hbm_stream.hip)


def _usable_cores():
    """threads this process can really run at once: the affinity mask, capped by a cgroup CPU quota when there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=50000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--genome", type=int, default=4641652)
    ap.add_argument("--node-len", type=int, default=64)
    ap.add_argument("--bandwidth", type=int, default=35)
    ap.add_argument("--graph", choices=["linear", "bubbles", "dense"], default="linear")
    ap.add_argument("--errors", default="0.04,0.04,0.04", help="substitution,insertion,deletion rates of the read simulator")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="reads timed through the CPU oracle (0 = skip)")
    ap.add_argument("--stamps", action="store_true", help="diagnostic build with per-phase cycle stamps (not a timed build)")
    ap.add_argument("--lib", default=None, help="alternative build of the library (experiments)")
    ap.add_argument("--accuracy", type=int, default=256, help="reads scored against the simulator's true paths with the reference's criterion (CompareAlignments.cpp)")
    ap.add_argument("--pipeline-chunks", type=int, default=12, help="chunks of the batch's size run through the overlapped host pipeline for detail.pipelined_host_to_host_Gbp_s (0 = skip)")
    ap.add_argument("--check", type=int, default=64, help="reads compared with the oracle after the run (includes failed / later-pass reads)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: ONE read set of --reads reads for the whole job, chunks pulled by the ranks from the shared queue "
                                                          "(sharding.align_queued); default is weak scaling, --reads per GPU")
    ap.add_argument("--kernel-only", action="store_true", help="time the device passes alone (profiling runs); the default step also brings the results to host memory")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the aligner has no CPU path")
    # GA_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get("GA_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
        # the host stages (job building, result assembly) are threaded: the ranks of one node share its cores
        os.environ.setdefault("GA_HOST_THREADS", str(max(1, _usable_cores() // world)))
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    import __graft_entry__ as entry
    # one rank builds (a fresh snapshot can make the in-tree library look stale); the others wait and only load it
    if world > 1:
        if rank == 0:
            entry.build_product()
        dist.barrier()
        os.environ["GA_SKIP_BUILD"] = "1"
    else:
        entry.build_product()
    from graphaligner_amd import binding, synth

    t0 = time.time()
    if args.graph == "linear":
        g = synth.linear_graph(args.genome, node_len=args.node_len, seed=42)
    elif args.graph == "bubbles":
        g = synth.bubble_graph(args.genome, node_len=args.node_len, seed=44)
    else:
        # chr22-like density: ~1 SNP per 45 bp, short indels, 32-bp nodes
        g = synth.SynthGraph(synth.random_genome(args.genome, 47), node_len=args.node_len, snp_every=45, indel_every=500, seed=48)
    e_sub, e_ins, e_del = (float(x) for x in args.errors.split(","))
    truth = []
    # (strong scaling: every rank holds the same read set, as every thread of the reference's driver does, Aligner.cpp:107-117)
    reads, seeds = synth.simulate_reads(g, args.reads, args.read_len, sub=e_sub, ins=e_ins, dele=e_del, seed=43 + (0 if args.strong else 1000 * rank), truth=truth)
    t_gen = time.time() - t0
    t0 = time.time()
    lib_path = entry.build_stamped() if args.stamps else args.lib
    if args.graph == "linear":
        # BASELINE's configuration is a single-contig GFA: ONE segment of the whole genome.  The reference would hand such a band to
        # its sparse method (>= 200 000 bp); the library's loader cuts the segment into node_len-bp pieces instead
        # (ga_graph_load_gfa_split), which is the same graph as the simulator's chain, ids included (the oracle below works on that)
        contig = "".join(seq for _, seq in g.nodes)
        graph = binding.Graph(gfa="H\tVN:Z:1.0\nS\t1\t%s\n" % contig, device=local, lib_path=lib_path, split=args.node_len)
        assert graph.node_count == 2 * len(g.nodes) + 2
        del contig
    else:
        graph = binding.Graph(gfa=g.gfa(), device=local, lib_path=lib_path)
    t_graph = time.time() - t0
    t0 = time.time()
    batch = graph.prepare(reads, seeds, args.bandwidth, 0)
    t_prep_batch = time.time() - t0
    total_bp = batch.total_bp

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # One step = one pass of the hot path over one batch whose reads, match words and job list are resident in HBM: every kernel pass
    # (band projection, fill, HMM stop test, traceback) AND the way back -- download of the traceback's node runs / moves and assembly
    # of the AlignmentResults in host memory (traceToAlignment / mergeAlignments).  --kernel-only leaves the way
    # back out (the figure round 2 printed as `value`; it stays in detail.kernel_only_Gbp_s and in the roofline).
    from graphaligner_amd import sharding
    # Consecutive steps overlap the way a caller's pipeline does (sharding.run_overlapped): while the host assembles step k's results,
    # the device runs step k + 1 -- two resident copies of the batch take turns, every step is complete (its results are in host memory)
    # before the clock stops.
    strong_chunk = max(64, min(65536, (args.reads + 4 * world - 1) // (4 * world)))
    from concurrent.futures import ThreadPoolExecutor
    twin = None if (args.kernel_only or args.strong) else graph.prepare(binding.ReadSet(reads, seeds), None, args.bandwidth, 0)
    kernel_ms, main_kernel_ms = [], []

    def run_steps(n, record):
        if args.strong and world > 1:
            for _ in range(n):
                sharding.align_queued(graph, reads, seeds, args.bandwidth, dist=dist, chunk_reads=strong_chunk, summary=True)
            return
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = None
            timeline = [] if (record and os.environ.get("GA_BENCH_TIMELINE")) else None
            t_base = time.perf_counter()

            def collect_of(b, k):
                ta = time.perf_counter()
                r = b.collect(True)
                if timeline is not None:
                    timeline.append(("collect", k, ta - t_base, time.perf_counter() - t_base))
                return r

            for k in range(n):
                b = batch if (twin is None or k % 2 == 0) else twin
                ta = time.perf_counter()
                b.run()
                if timeline is not None:
                    timeline.append(("run", k, ta - t_base, time.perf_counter() - t_base))
                if record:
                    sk = b.stats()
                    kernel_ms.append(sk["kernel_ms"])
                    main_kernel_ms.append(sk["main_kernel_ms"])
                if pending is not None:
                    pending.result()
                pending = None if args.kernel_only else pool.submit(collect_of, b, k)
            if pending is not None:
                pending.result()
            if timeline:
                for what, k, ta, tb in sorted(timeline, key=lambda e: e[2]):
                    print("   %-8s step %d  %7.1f .. %7.1f ms" % (what, k, ta * 1e3, tb * 1e3), file=sys.stderr)

    run_steps(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    if not kernel_ms:
        batch.run()
        sk = batch.stats()
        kernel_ms.append(sk["kernel_ms"])
        main_kernel_ms.append(sk["main_kernel_ms"])
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    del twin
    # what was aligned (reads returned with failed = 0), from one collect after the timed region
    t0 = time.time()
    summary = batch.collect(summary=True)
    t_collect = time.time() - t0
    st = batch.stats()
    # the host stages once more on a second batch of the same reads: the first prepare / collect of a process also pay for the pinned
    # download buffer and the first touch of their arrays (the PCIe-inclusive figure in detail is the better of the two)
    if rank == 0 and args.pipeline_chunks > 0:       # (--pipeline-chunks 0: no launches beyond the timed ones, for profiling runs)
        rs2 = binding.ReadSet(reads, seeds)           # (the arrays a C caller holds; the Python lists are the simulator's)
        t0 = time.time()
        batch2 = graph.prepare(rs2, None, args.bandwidth, 0)
        t1 = time.time()
        batch2.run()
        t2 = time.time()
        batch2.collect(summary=True)
        t3 = time.time()
        if (t1 - t0) + (t3 - t2) < t_prep_batch + t_collect:
            t_prep_batch, t_collect = t1 - t0, t3 - t2
        del batch2
    # ... and the three stages overlapped (sharding.run_overlapped: job building + upload of chunk k+1, kernels of chunk k, download +
    # assembly of chunk k-1 on separate host threads and streams) over chunks of this batch's size.  The reads are handed over as the
    # C ABI takes them (binding.ReadSet, built outside the clock: a caller in C or C++ holds such arrays already)
    t_pipe = None
    if rank == 0 and args.pipeline_chunks > 0:
        from graphaligner_amd import sharding
        rs = binding.ReadSet(reads, seeds)
        t0 = time.time()
        got = sharding.run_overlapped(graph, ((k, rs) for k in range(args.pipeline_chunks)), args.bandwidth, summary=True)
        t_pipe = time.time() - t0
        assert len(got) == args.pipeline_chunks and all(int((r["failed"] == 0).sum()) == int((summary["failed"] == 0).sum()) for _, r in got)
        del got
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    aligned_bp = int(lens[summary["failed"] == 0].sum())
    n_failed = int((summary["failed"] != 0).sum())
    if world > 1 and not args.strong:
        t = torch.tensor([float(aligned_bp)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        aligned_total = float(t.item())
    else:
        aligned_total = float(aligned_bp)          # (strong scaling: the one read set, whichever rank aligned which chunk)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = aligned_total / (elapsed / args.steps) / 1e9
    k_ms = float(np.mean(kernel_ms))                   # all kernel passes of a step
    main_ms = float(np.mean(main_kernel_ms))           # the dominant kernel alone: the first pass over all jobs
    variant = int(st["main_variant"])
    kernel_name = ("ga_lanes_kernel<%d,%d>" % (variant // 1000, 32 if variant % 10 else 64)) if variant > 0 else "ga_extend_kernel<%d,false>" % (-variant if variant else 64)
    # the roofline prices the dominant kernel: the first pass when it finished every job; when jobs went on to the ladder their column
    # updates were (re)computed by later passes, and the time of ALL passes is what the batch's column updates are divided by
    dom_ms = main_ms if (variant and st["jobs_retried"] == 0) else k_ms
    achieved = BYTES_PER_COLUMN_UPDATE * st["column_updates"] / (dom_ms * 1e-3) / 1e9
    # HBM traffic per launch: PMC counters cannot be read from inside this process, so the per-column-update figure measured with
    # rocprofv3 --pmc on this kernel and this workload (tools/pmc_lanes.sh -> profiles/r2_hbm_traffic.json) is scaled to this launch
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "r3_hbm_traffic.json")
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "r2_hbm_traffic.json")
    if os.path.exists(tpath) and args.graph == "linear":
        tj = json.load(open(tpath))
        if tj.get("kernel", "").startswith(kernel_name.split("<")[0]):
            traffic = int(tj["hbm_bytes_per_column_update"] * st["column_updates"])
            traffic_note = "scaled from profiles/%s (%s B per column update, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of build %s)" % (os.path.basename(tpath), tj["hbm_bytes_per_column_update"], tj.get("build", "?"))
    if args.graph == "linear":
        workload = ("E. coli-scale single-contig GFA (one %d-bp segment, seed 42, loaded cut into %d-bp pieces) + %d x %d bp simulated ONT-error reads (s,i,d=%s, seed 43), band=%d, 1 seed/read at pos 0"
                    % (args.genome, args.node_len, args.reads, args.read_len, args.errors, args.bandwidth))
    elif args.graph == "bubbles":
        workload = ("yeast-like pangenome GFA (%d bp, SNP / indel / SV bubbles, nodes <= %d bp, seed 44) + %d x %d bp reads (s,i,d=%s), band=%d"
                    % (args.genome, args.node_len, args.reads, args.read_len, args.errors, args.bandwidth))
    else:
        workload = ("chr22-like dense graph (%d bp seed 47, a SNP every ~45 bp and short indels seed 48, nodes <= %d bp) + %d x %d bp reads (s,i,d=%s), band=%d"
                    % (args.genome, args.node_len, args.reads, args.read_len, args.errors, args.bandwidth))
    e2e_s = t_prep_batch + k_ms * 1e-3 + t_collect
    out = {
        "metric": "aligned Gbp/sec (whole node), 10kb ONT reads vs chr-scale GFA",
        "value": round(value, 4), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload, ("reads_total" if args.strong else "reads_per_gpu"): args.reads, "read_len": args.read_len, "bandwidth": args.bandwidth, "graph_bp_both_strands": int(graph.bp),
                   "step": ("device passes only" if args.kernel_only else "reads / match words / jobs resident in HBM -> every kernel pass -> AlignmentResults assembled in host memory; "
                                                                            "the host assembles step k while the device runs step k + 1"),
                   "parallelism": ("one read set, chunks of %d reads pulled from a shared queue, graph replicated, no collective" % strong_chunk) if args.strong else "reads sharded, graph replicated, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                     "traffic": traffic, "traffic_source": traffic_note, "kernel": kernel_name, "kernel_ms": round(dom_ms, 3), "frac_of_measured_stream_copy": round(achieved / HBM_STREAM_GBS, 5),
                     "column_updates_per_launch": int(st["column_updates"]), "bytes_per_column_update": BYTES_PER_COLUMN_UPDATE},
        "detail": {"reads_failed": n_failed, "jobs": int(st["n_jobs"]), "jobs_left_to_the_wave_per_read_ladder": int(st["jobs_retried"]), "all_passes_ms": round(k_ms, 3),
                   "waves": int(st["slots"]), "scratch_GB": round(st["scratch_bytes"] / 1e9, 2), "gen_s": round(t_gen, 1), "graph_upload_s": round(t_graph, 2),
                   "prepare_s": round(t_prep_batch, 2), "collect_s": round(t_collect, 2),
                   "pipelined_host_to_host_Gbp_s": (round(total_bp * args.pipeline_chunks / t_pipe / 1e9, 3) if t_pipe else None),
                   # reads in host memory -> results in host memory: job building + upload, all kernel passes, download + assembly
                   "end_to_end_host_to_host_Gbp_s": round(aligned_bp / e2e_s / 1e9, 3),
                   "kernel_only_Gbp_s": round(aligned_bp / (k_ms * 1e-3) / 1e9, 4),
                   "G_column_updates_per_s": round(st["column_updates"] / (dom_ms * 1e-3) / 1e9, 3), "GCUPS": round(64 * st["column_updates"] / (dom_ms * 1e-3) / 1e9, 1)},
    }

    # ---- CPU baseline: the oracle (a port of the reference algorithm), host cores, bounded sample ----
    if args.cpu_sample > 0 and world == 1:            # (on rank 0 of a single-GPU run only)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob
        cores = _usable_cores()
        n = min(args.cpu_sample, len(reads))
        og = ob.OracleGraph(g.nodes, g.edges)
        b = og.bench(reads[:n], seeds[:n], args.bandwidth, 0, cores)
        n1 = max(1, min(n, args.cpu_sample // 16))
        b1 = og.bench(reads[:n1], seeds[:n1], args.bandwidth, 0, 1)
        out["cpu_baseline"] = {"value": round(b["aligned_bp"] / b["seconds"] / 1e9, 6), "unit": "Gbp/s", "cores": cores, "kind": "port",
                               "value_1_thread": round(b1["aligned_bp"] / b1["seconds"] / 1e9, 6),
                               "sample": "first %d reads of the same batch, %d threads popping reads from a shared queue (Aligner.cpp:285-298), %.1f s; 1 thread: first %d reads, %.1f s"
                                         % (n, cores, b["seconds"], n1, b1["seconds"])}
    # spot check against the oracle: a regular sample plus every read that failed or was not finished by the first pass
    if args.check > 0 and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob
        import parity_common as pc
        og = ob.OracleGraph(g.nodes, g.edges)
        k = min(args.check, len(reads))
        if k:
            special = [int(i) for i in np.nonzero((summary["failed"] != 0) | (summary["status"] != 0) | (summary["reserved"] != 0))[0][:k // 2]]
            step = max(1, len(reads) // max(1, k - len(special)))
            pick = sorted(set(list(range(0, len(reads), step))[:k - len(special)] + special))
            some = graph.align([reads[i] for i in pick], [seeds[i] for i in pick], args.bandwidth, 0)
            for d, i in zip(some, pick):
                assert d["score"] == int(summary["score"][i]) or d["failed"]
                pc.compare_read(dict(d, trace=np.zeros((0, 7), dtype=np.int64)), dict(og.align(reads[i], [seeds[i]], args.bandwidth), trace=np.zeros((0, 7), dtype=np.int64)), "bench read %d" % i)
            out["detail"]["oracle_spot_check_reads"] = len(pick)
            out["detail"]["oracle_spot_check_failed_or_later_pass_reads"] = len(special)
    if args.accuracy > 0:
        # node-set overlap of predicted vs true path, good when >= 0.7 (CompareAlignments.cpp:13-44, 86)
        from graphaligner_amd import compare
        k = min(args.accuracy, len(reads))
        some = graph.align(reads[:k], seeds[:k], args.bandwidth, 0)
        sizes = {nid: len(seq) for nid, seq in g.nodes}
        rep = compare.compare({"r%d" % i: truth[i] for i in range(k)}, {"r%d" % i: compare.predicted_nodes(r) for i, r in enumerate(some) if not r["failed"]}, sizes)
        out["detail"]["accuracy"] = {"reads": k, "good_matches": rep["good"], "bad_matches": rep["bad"], "criterion": "node-set overlap >= 0.7 (CompareAlignments.cpp:86)"}
    if args.stamps:
        names = ["end_slice", "band+order", "trace_fast(in traceback)", "trace_general(in traceback)", "fill", "traceback", "trace_handover(in traceback)", "rounds|fast_iterations<<32"]
        if os.environ.get("GA_STAMPS_LEVEL") == "2":      # the traceback's general step in parts instead of the slice phases
            names[0], names[1], names[4] = "general:decide", "general:slice_change", "general:window"
        if os.environ.get("GA_STAMPS_LEVEL") == "4":      # the band phase in parts; [6] = the band phase, [7] = the fill
            names = ["band:map_order", "band:previous_band", "band:heap", "band:slots", "band:processing_order", "traceback", "band phase", "fill"]
        if os.environ.get("GA_STAMPS_LEVEL") in ("3", "4"):
            tot = float(st["stamps"][5] + st["stamps"][6] + st["stamps"][7]) or 1.0
        else:
            tot = float(st["stamps"][0] + st["stamps"][1] + st["stamps"][4] + st["stamps"][5]) or 1.0
        out["detail"]["phase_share"] = {n: round(v / tot, 4) for n, v in zip(names, st["stamps"])}
        out["detail"]["cycles_per_job"] = round(tot / max(1, st["n_jobs"]))
        out["detail"]["stamps_raw"] = [int(v) for v in st["stamps"]]
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
