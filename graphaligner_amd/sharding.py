"""Read sharding across the GPUs of one node.

Reads are independent (the reference's driver hands them to threads one by one,
Aligner.cpp:107-117) and the graph is read-only, so the multi-GPU form is: one process per GPU,
the graph replicated in every GPU's HBM, reads dealt longest-first round-robin, results put back
in the original order on rank 0.  There is no collective on the data path; torch.distributed is
only used to gather the per-read results (and, in bench.py, to take the max time over ranks)."""
import numpy as np


def shard_indices(lengths, rank, world):
    """indices of the reads rank `rank` aligns: sort by length (longest first, stable), deal round-robin"""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    return [int(i) for i in order[rank::world]]


def align_sharded(graph, reads, seeds, bandwidth, ramp=0, flags=0, dist=None):
    """every rank calls this with the SAME reads/seeds and its own `graph` (already uploaded to its
    GPU); rank 0 gets the full result list in input order, other ranks get None"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return graph.align(reads, seeds, bandwidth, ramp, flags)
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_indices([len(r) for r in reads], rank, world)
    local = graph.align([reads[i] for i in mine], [seeds[i] for i in mine], bandwidth, ramp, flags) if mine else []
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, local))
    if rank != 0:
        return None
    out = [None] * len(reads)
    for idx, res in gathered:
        for i, r in zip(idx, res):
            out[i] = r
    return out


# ---- work queue (the reference's scheduler is a mutex-guarded pop, Aligner.cpp:107-117, 285-298) ----------------
def make_chunks(lengths, chunk_reads):
    """reads sorted longest first, cut into chunks of `chunk_reads` (a chunk should fill a GPU: one lanes = reads wave carries 64
    reads and an MI355X holds 1024 such waves, so the default is 65536)"""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    return [[int(i) for i in order[k:k + chunk_reads]] for k in range(0, len(order), chunk_reads)]


_queue_calls = 0       # how many queues this process has drawn from: every rank calls align_queued the same number of times, so the
                       # number names the same queue on all of them


class _Counter:
    """the shared "next chunk" counter: the process group's store (a TCP store on 127.0.0.1 for a single node) when there are
    several ranks, a local integer otherwise.  Every queue gets a key of its own (tag + the call's sequence number): a store key is
    never reset, so a second queue under the first one's key would find it drained."""

    def __init__(self, dist, tag):
        global _queue_calls
        _queue_calls += 1
        self.key, self.local, self.store = "%s/%d" % (tag, _queue_calls), 0, None
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed import distributed_c10d as c10d
            self.store = c10d._get_default_store()

    def next(self):
        if self.store is None:
            self.local += 1
            return self.local - 1
        return int(self.store.add(self.key, 1)) - 1


def run_overlapped(graph, inputs, bandwidth, ramp=0, flags=0, summary=False, gam=False):
    """`inputs` yields (key, (reads, seeds)) or (key, binding.ReadSet); three stages overlap on separate host threads and HIP streams:
    job building + upload of input k+1, the kernels of input k, download + assembly of input k-1.  Returns [(key, results)]; results
    are Python lists, per-read record arrays (summary=True) or the bytes of one GAM group per input (gam=True)."""
    from concurrent.futures import ThreadPoolExecutor
    it = iter(inputs)

    def prep():
        try:
            key, what = next(it)
        except StopIteration:
            return None
        if isinstance(what, tuple):
            return key, graph.prepare(what[0], what[1], bandwidth, ramp, flags)
        return key, graph.prepare(what, None, bandwidth, ramp, flags)

    def finish(batch):
        # (the batch is freed as soon as its results are out: its device buffers and its pinned copy of the reads serve the next one)
        try:
            return batch.collect_gam() if gam else batch.collect(summary)
        finally:
            batch.close()

    done = []                         # (key, future of its results)
    with ThreadPoolExecutor(max_workers=2) as pool:
        nxt = pool.submit(prep)
        while True:
            got = nxt.result()
            if got is None:
                break
            key, batch = got
            nxt = pool.submit(prep)                                      # built and uploaded while this input runs
            batch.run()
            done.append((key, pool.submit(finish, batch)))                # assembled while the next input runs
        return [(k, f.result()) for k, f in done]


def align_queued(graph, reads, seeds, bandwidth, ramp=0, flags=0, dist=None, chunk_reads=65536, tag="ga_queue", summary=False, gam=False, names=None):
    """every rank calls this with the SAME reads/seeds and its own `graph` (already uploaded to its GPU).  Chunks of reads are pulled
    from a shared counter, so a rank that finishes early takes more; on each rank the stages of consecutive chunks overlap
    (run_overlapped).  Rank 0 gets the full result in input order, other ranks get None.  No collective on the data path; what travels
    to rank 0 at the end is, per chunk,
      summary=True  a numpy record array (Batch.collect(summary=True)), merged into one array in input order;
      gam=True      the bytes of a GAM group (ga_results_encode_gam): rank 0 returns them joined in chunk order -- GAM groups
                    concatenate (stream.hpp:24-63), so that is the GAM file of the whole read set, reads ordered longest first;
      otherwise     Python lists of per-read dicts (tests and small inputs: pickling 200 000 of those is not a transport)."""
    chunks = make_chunks([len(r) for r in reads], chunk_reads)
    counter = _Counter(dist, tag)
    multi = counter.store is not None

    def taken():
        while True:
            k = counter.next()
            if k >= len(chunks):
                return
            idx = chunks[k]
            if gam and names is not None:
                from . import binding
                yield k, binding.ReadSet([reads[i] for i in idx], [seeds[i] for i in idx], [names[i] for i in idx])
            else:
                yield k, ([reads[i] for i in idx], [seeds[i] for i in idx])

    mine = run_overlapped(graph, taken(), bandwidth, ramp, flags, summary, gam)
    if multi:
        gathered = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if dist.get_rank() != 0:
            return None
        mine = [kr for part in gathered for kr in part]
    got = sorted(k for k, _ in mine)
    if got != list(range(len(chunks))):
        raise RuntimeError("work queue %s: chunks %s came back for %d chunks" % (counter.key, got[:20], len(chunks)))
    if gam:
        return b"".join(res for _, res in sorted(mine, key=lambda kr: kr[0]))
    if summary:
        out = np.zeros(len(reads), dtype=mine[0][1].dtype) if mine else np.zeros(0)
        for k, res in mine:
            out[np.asarray(chunks[k], dtype=np.int64)] = res
        return out
    out = [None] * len(reads)
    for k, res in mine:
        for i, r in zip(chunks[k], res):
            out[i] = r
    return out
