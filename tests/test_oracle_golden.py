"""Oracle against the committed fixtures (tests/golden/, built by tools/make_golden.py).

What is pinned by the REFERENCE here is thin, because the reference ships data files but no
expected outputs and its engine cannot be built in this image:
  * the two observations SURVEY.md section 8(c) recorded while the real engine was run during the
    survey: test/gwws_fail_ex1.vg + its 290 bp longest-path read aligns with score 30, and
    test/smallexample as shipped dies on assert(slice.samplingFrequency > 1) (GraphAligner.h:906).
The oracle_vectors.json cases are regression pins produced by the oracle itself (parity unpinned);
they are also replayed through the emulated device program here and through the GPU in
test_gpu_parity.py::test_golden_vectors."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
import parity_common as pc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def test_reference_fixture_gwws_longest_path_scores_30():
    d = _load("ref_gwws_fail_ex1.json")
    assert len(d["nodes"]) == 20 and len(d["edges"]) == 27 and len(d["longest_path_read"]) == 290
    og = ob.OracleGraph([tuple(x) for x in d["nodes"]], [tuple(x) for x in d["edges"]])
    r = og.align(d["longest_path_read"], [(d["longest_path"][0], 0, False)], 35)
    # 290 bp pads to 320 rows; the 30 padded N rows run past the end of the graph: 30 insertions
    assert r["status"] == 0 and not r["failed"] and r["score"] == 30


def test_reference_fixture_smallexample_fails_like_the_reference():
    d = _load("ref_smallexample.json")
    assert len(d["nodes"]) == 18 and len(d["edges"]) == 25 and len(d["read"]) == 66
    og = ob.OracleGraph([tuple(x) for x in d["nodes"]], [tuple(x) for x in d["edges"]])
    s = d["seeds"][0]
    assert (s["node"], s["pos"], s["reverse"]) == (6738, 0, 0)
    r = og.align(d["read"], [(s["node"], s["pos"], bool(s["reverse"]))], 35)
    assert r["status"] == 1 and r["failed"] and "samplingFrequency" in r["message"]


def _check_case(case, results):
    for i, (exp, got) in enumerate(zip(case["expected"], results)):
        ctx = "%s read %d" % (case["name"], i)
        assert got["status"] == exp["status"] and got["failed"] == exp["failed"], ctx
        if exp["failed"]:
            continue
        for k in ("score", "query_position", "alignment_start", "alignment_end", "columns"):
            assert got[k] == exp[k], (ctx, k, got[k], exp[k])
        assert [list(m) for m in got["mappings"]] == exp["mappings"], ctx
        assert got["trace"].shape[0] == exp["n_trace"], ctx
        assert int(np.asarray(got["trace"], dtype=np.int64).sum() % (1 << 61)) == exp["trace_checksum"], ctx


def test_oracle_reproduces_golden_vectors():
    for case in _load("oracle_vectors.json")["cases"]:
        og = ob.OracleGraph([tuple(x) for x in case["nodes"]], [tuple(x) for x in case["edges"]])
        res = [og.align(r, [tuple(s)], case["bandwidth"], case.get("ramp", 0)) for r, s in zip(case["reads"], case["seeds"])]
        _check_case(case, res)


def run_device_on_golden(lib_path):
    from graphaligner_amd import binding
    for case in _load("oracle_vectors.json")["cases"]:
        g = binding.Graph([tuple(x) for x in case["nodes"]], [tuple(x) for x in case["edges"]], lib_path=lib_path)
        res = g.align(case["reads"], [tuple(s) for s in case["seeds"]], case["bandwidth"], case.get("ramp", 0), flags=binding.GA_F_TRACE)
        _check_case(case, res)
    for name, seedf in (("ref_gwws_fail_ex1.json", None), ("ref_smallexample.json", None)):
        d = _load(name)
        g = binding.Graph([tuple(x) for x in d["nodes"]], [tuple(x) for x in d["edges"]], lib_path=lib_path)
        if "longest_path_read" in d:
            r = g.align([d["longest_path_read"]], [(d["longest_path"][0], 0, False)], 35)[0]
            assert r["status"] == 0 and r["score"] == 30
        else:
            s = d["seeds"][0]
            r = g.align([d["read"]], [(s["node"], s["pos"], bool(s["reverse"]))], 35)[0]
            assert r["status"] == 1 and r["failed"]


def test_emulated_device_program_reproduces_golden_vectors():
    run_device_on_golden(pc.emul_lib_path())
