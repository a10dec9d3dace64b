"""The device program (graphaligner_amd/csrc/ga_kernel.h) executed on the host with every wave64
primitive emulated (tests/emul), checked against the CPU oracle.  This is how the device LOGIC
is kept honest in the GPU-less build container; the `gpu` tests repeat the same cases through
the real HIP library.  The emulation is test infrastructure and is never shipped."""
import pytest

import parity_cases as cases
import parity_common as pc


@pytest.fixture(scope="module")
def lib():
    return pc.emul_lib_path()


def test_linear(lib):
    cases.case_wave_primitives_on_hardware(lib)


@pytest.mark.parametrize("node_len,snp,indel,sv", cases.RANDOM_GRAPHS)
def test_random_graphs(lib, node_len, snp, indel, sv):
    cases.case_random_graphs(node_len, snp, indel, sv, lib)


@pytest.mark.parametrize("node_len,bw,back_edges,self_loops,max_span", cases.CYCLIC_GRAPHS)
def test_cyclic_graphs(lib, node_len, bw, back_edges, self_loops, max_span):
    cases.case_cyclic_graphs(node_len, bw, back_edges, self_loops, max_span, lib)


@pytest.mark.parametrize("node_len,bw,ramp,err", cases.RAMP_CASES)
def test_ramp_redo(lib, node_len, bw, ramp, err):
    cases.case_ramp_redo(node_len, bw, ramp, err, lib)


def test_gfa_overlap(lib):
    cases.case_gfa_overlap(lib)


def test_inversion_edges(lib):
    cases.case_inversion_edges(lib)


def test_reference_graph_plumbing(lib):
    cases.case_reference_graph_plumbing(lib)


def test_short_and_edge_reads(lib):
    cases.case_short_and_edge_reads(lib)


def test_iupac_n_and_invalid_characters(lib):
    cases.case_iupac_n_and_invalid_characters(lib)


def test_multiple_seeds_per_read(lib):
    cases.case_multiple_seeds_per_read(lib)


def test_results_without_trace_items(lib):
    cases.case_results_without_trace_items(lib)


def test_unknown_seed_node_reports_bad_seed(lib):
    cases.case_unknown_seed_node_reports_bad_seed(lib)


def test_gfa_loader_matches_node_edge_api(lib):
    cases.case_gfa_loader_matches_node_edge_api(lib)


@pytest.mark.parametrize("branches,branch_len,shared,stem,bw,ramp", cases.SPARSE_FANS)
def test_sparse_method_and_override(lib, branches, branch_len, shared, stem, bw, ramp):
    cases.case_sparse_method_and_override(branches, branch_len, shared, stem, bw, ramp, lib)


def test_sparse_sharp_edges(lib):
    cases.case_sparse_sharp_edges(lib)


def test_batch_run_twice(lib):
    cases.case_batch_run_twice(lib)


def test_trace_pool_overflow(lib):
    cases.case_trace_pool_overflow(lib)
