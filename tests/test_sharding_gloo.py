"""The N>1 path: one process per GPU, reads sharded, graph replicated, no data-path collective.
Rehearsed here on CPU with world_size 2 over gloo; each rank drives the host emulation of the
device program (there is no GPU in this container), rank 0 compares the merged results with a
single-process run and with the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lib, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from graphaligner_amd import binding, sharding, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = synth.bubble_graph(20000, node_len=32, seed=31)
    reads, seeds = synth.simulate_reads(g, 10, 900, seed=32)
    reads = [r[: 500 + 40 * i] for i, r in enumerate(reads)]       # ragged lengths
    graph = binding.Graph(g.nodes, g.edges, lib_path=lib)
    reads[3] = reads[3][:120]                                       # fails (fewer than four slices): the failure must come back in place
    res = sharding.align_sharded(graph, reads, seeds, 35, dist=dist)
    # the work queue: chunks of 3 reads pulled from the shared counter (ragged: 10 reads = 3 + 3 + 3 + 1), whichever rank is free
    queued = sharding.align_queued(graph, reads, seeds, 35, dist=dist, chunk_reads=3)
    if rank == 0:
        single = graph.align(reads, seeds, 35)
        same = lambda x, y: all(a["score"] == b["score"] and a["mappings"] == b["mappings"] and a["status"] == b["status"] and a["failed"] == b["failed"] for a, b in zip(x, y))
        ok = same(res, single) and same(queued, single) and single[3]["failed"] and sum(1 for r in single if r["failed"]) == 1
        import oracle_binding as ob
        og = ob.OracleGraph(g.nodes, g.edges)
        ok2 = all(r["score"] == og.align(x, [s], 35)["score"] for r, x, s in zip(res, reads, seeds) if not r["failed"])
        q.put((ok, ok2, len(res)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_reads_and_merge_in_order():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_common as pc
    lib = pc.emul_lib_path()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lib, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ok, ok2, n = q.get(timeout=10)
    assert ok and ok2 and n == 10


def _worker_queue(rank, world, port, lib, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GA_HOST_THREADS"] = "2"                              # (cores / ranks, as bench.py sets it)
    import torch.distributed as dist
    from graphaligner_amd import binding, sharding, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = synth.bubble_graph(30000, node_len=32, seed=33)
    reads, seeds = synth.simulate_reads(g, 1000, 420, seed=34)
    reads = [r[: 280 + (37 * i) % 140] for i, r in enumerate(reads)]
    names = ["r%04d" % i for i in range(len(reads))]
    graph = binding.Graph(g.nodes, g.edges, lib_path=lib)
    # the queue twice in one process group (a store key is never reset: each call must get its own), three transports
    first = sharding.align_queued(graph, reads, seeds, 35, dist=dist, chunk_reads=64, summary=True)
    gam = sharding.align_queued(graph, reads, seeds, 35, dist=dist, chunk_reads=64, gam=True, names=names)
    lists = sharding.align_queued(graph, reads[:100], seeds[:100], 35, dist=dist, chunk_reads=16)
    if rank == 0:
        alone_summary = sharding.align_queued(graph, reads, seeds, 35, chunk_reads=64, summary=True)
        alone_gam = sharding.align_queued(graph, reads, seeds, 35, chunk_reads=64, gam=True, names=names)
        alone = graph.align(reads[:100], seeds[:100], 35)
        same_lists = all(a["score"] == b["score"] and a["mappings"] == b["mappings"] and a["status"] == b["status"] for a, b in zip(lists, alone))
        ok_summary = all((first[f] == alone_summary[f]).all() for f in ("status", "failed", "score", "n_mappings", "alignment_start", "alignment_end", "query_position", "column_updates"))
        n_ok = int((first["failed"] == 0).sum())
        q.put((ok_summary, gam == alone_gam, same_lists, n_ok, len(gam)))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_pull_1000_reads_from_the_queue_twice():
    """world_size 4: 1 000 reads in chunks of 64 from the shared counter, called three times in one process group; per-read records and
    the GAM bytes gathered on rank 0 are identical to a single process's"""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_common as pc
    lib = pc.emul_lib_path()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_queue, args=(r, 4, port, lib, q)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    ok_summary, same_gam, same_lists, n_ok, gam_len = q.get(timeout=10)
    assert ok_summary and same_gam and same_lists
    assert n_ok >= 900 and gam_len > 50000


def test_shard_indices_cover_everything_once():
    from graphaligner_amd import sharding
    lengths = np.random.default_rng(1).integers(100, 20000, size=101)
    seen = []
    for r in range(8):
        idx = sharding.shard_indices(lengths, r, 8)
        seen += idx
        # longest first within a shard
        assert all(lengths[a] >= lengths[b] for a, b in zip(idx, idx[1:]))
    assert sorted(seen) == list(range(101))
