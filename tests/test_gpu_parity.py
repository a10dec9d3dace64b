"""GPU parity tests: the real library (graphaligner_amd/libgraphaligner_amd.so, HIP, gfx950)
through the C ABI against the CPU oracle on identical seeded inputs.  Bit-exact: status, score,
node path, per-node edits (from/to length + sequence), offsets, strands, query position,
alignment start/end, TraceItem lists, and the column-update count.  Cases: parity_cases.py."""
import pytest

import parity_cases as cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need a real MI355X"


@pytest.fixture(autouse=True, params=["lanes-first", "by-graph-shape"])
def _first_pass(request, monkeypatch):
    """every case runs twice: with the lanes = reads kernel forced as the first pass (whatever the graph looks like; jobs it declines
    climb the wave-per-read ladder), and with the library's own choice by graph shape (short-node graphs start in the wave-per-read kernel)"""
    if request.param == "lanes-first":
        monkeypatch.setenv("GA_LANES", "1")
    else:
        monkeypatch.delenv("GA_LANES", raising=False)


def test_wave_primitives_on_hardware():
    cases.case_wave_primitives_on_hardware()


@pytest.mark.parametrize("node_len,snp,indel,sv", cases.RANDOM_GRAPHS)
def test_random_graphs(node_len, snp, indel, sv):
    cases.case_random_graphs(node_len, snp, indel, sv)


@pytest.mark.parametrize("node_len,bw,back_edges,self_loops,max_span", cases.CYCLIC_GRAPHS)
def test_cyclic_graphs(node_len, bw, back_edges, self_loops, max_span):
    cases.case_cyclic_graphs(node_len, bw, back_edges, self_loops, max_span)


@pytest.mark.parametrize("node_len,bw,ramp,err", cases.RAMP_CASES)
def test_ramp_redo(node_len, bw, ramp, err):
    cases.case_ramp_redo(node_len, bw, ramp, err)


def test_gfa_overlap():
    cases.case_gfa_overlap()


def test_inversion_edges():
    cases.case_inversion_edges()


def test_reference_graph_plumbing():
    cases.case_reference_graph_plumbing()


def test_short_and_edge_reads():
    cases.case_short_and_edge_reads()


def test_iupac_n_and_invalid_characters():
    cases.case_iupac_n_and_invalid_characters()


def test_multiple_seeds_per_read():
    cases.case_multiple_seeds_per_read()


def test_results_without_trace_items():
    cases.case_results_without_trace_items()


def test_unknown_seed_node_reports_bad_seed():
    cases.case_unknown_seed_node_reports_bad_seed()


def test_gfa_loader_matches_node_edge_api():
    cases.case_gfa_loader_matches_node_edge_api()


@pytest.mark.parametrize("branches,branch_len,shared,stem,bw,ramp", cases.SPARSE_FANS)
def test_sparse_method_and_override(branches, branch_len, shared, stem, bw, ramp):
    """bands of >= 200 000 cells: calculateSliceAlternate and the backtrace override on the device (ga_sparse.h)"""
    devs, oras = cases.case_sparse_method_and_override(branches, branch_len, shared, stem, bw, ramp)
    # (such jobs are finished by the last kernel of the ladder, never by the first pass)
    assert all(d["kernel_pass"] > 0 for d, o in zip(devs, oras) if o["sparse_slices"] > 0)


def test_sparse_sharp_edges():
    cases.case_sparse_sharp_edges()


def test_full_size_properties():
    cases.case_full_size_properties()


def test_golden_vectors():
    import test_oracle_golden as tg
    tg.run_device_on_golden(None)


def test_randomised_campaign():
    """a few hundred reads over random graph shapes, bandwidths, error rates, seed positions"""
    import subprocess, sys, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "parity_campaign.py"), "--trials", "30", "--reads", "16", "--seed", "5"],
                         capture_output=True, text=True, timeout=600)
    stats = json.loads(out.stdout.strip().split("\n")[-1])
    assert stats["mismatches"] == 0, stats
    assert stats["compared"] >= 300


def test_long_reads_on_short_nodes():
    cases.case_long_reads_on_short_nodes()


def test_batch_run_twice():
    cases.case_batch_run_twice()


def test_trace_pool_overflow():
    cases.case_trace_pool_overflow()
