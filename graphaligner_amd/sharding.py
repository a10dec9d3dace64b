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
