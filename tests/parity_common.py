"""shared helpers for the parity tests: run the same reads through the C ABI (real GPU library
or the host emulation of the device program) and through the CPU oracle, then compare every
field the reference's AlignmentResult carries."""
import os
import subprocess

import numpy as np

from graphaligner_amd import binding, synth
import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_SO = os.path.join(ROOT, "tests", "_build", "libga_emul.so")

# oracle status -> ABI status
_STATUS_MAP = {0: 0, 1: 1, 2: 2, 3: 3}


def emul_lib_path():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul")])
    return EMUL_SO


def compare_read(dev, ora, ctx=""):
    """dev: dict from graphaligner_amd.binding; ora: dict from oracle_binding"""
    assert dev["status"] == _STATUS_MAP[ora["status"]], (ctx, "status", dev["status"], ora["status"], ora["message"])
    assert dev["failed"] == ora["failed"], (ctx, "failed")
    if ora["status"] != 0 or ora["failed"]:
        return
    assert dev["score"] == ora["score"], (ctx, "score", dev["score"], ora["score"])
    assert dev["query_position"] == ora["query_position"], (ctx, "query_position")
    assert dev["alignment_start"] == ora["alignment_start"], (ctx, "alignment_start")
    assert dev["alignment_end"] == ora["alignment_end"], (ctx, "alignment_end")
    assert len(dev["mappings"]) == len(ora["mappings"]), (ctx, "n mappings", len(dev["mappings"]), len(ora["mappings"]))
    for i, (a, b) in enumerate(zip(dev["mappings"], ora["mappings"])):
        assert tuple(a) == tuple(b), (ctx, "mapping", i, a, b)
    if dev["trace"].shape[0] or ora["trace"].shape[0]:
        assert dev["trace"].shape == ora["trace"].shape, (ctx, "trace items", dev["trace"].shape, ora["trace"].shape)
        assert (dev["trace"] == ora["trace"]).all(), (ctx, "trace items differ")
    assert dev["columns"] == ora["columns"], (ctx, "column updates", dev["columns"], ora["columns"])


def run_both(nodes, edges, reads, seeds, bw, ramp=0, overlap=0, lib_path=None, trace=True):
    og = ob.OracleGraph(nodes, edges, overlap=overlap)
    g = binding.Graph(nodes, edges, overlap=overlap, lib_path=lib_path)
    devs = g.align(reads, seeds, bw, ramp, flags=binding.GA_F_TRACE if trace else 0)
    oras = []
    for r, s in zip(reads, seeds):
        lst = [s] if (len(s) == 3 and not isinstance(s[0], (tuple, list))) else list(s)
        oras.append(og.align(r, lst, bw, ramp))
    return devs, oras


def check_parity(nodes, edges, reads, seeds, bw, ramp=0, overlap=0, lib_path=None, ctx=""):
    devs, oras = run_both(nodes, edges, reads, seeds, bw, ramp, overlap, lib_path)
    for i, (d, o) in enumerate(zip(devs, oras)):
        compare_read(d, o, "%s read %d seed %s" % (ctx, i, seeds[i]))
    return devs, oras
