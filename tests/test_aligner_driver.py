"""The driver mirror (graphaligner_amd/aligner.py; reference: Aligner.cpp:107-322, AlignerMain.cpp) run end to end
on files: GFA + fastq + seed GAM in, GAM + per-read files out.  CPU tests run the device program through the
host emulation (tests/emul); the `gpu` test runs the same through the real library."""
import gzip
import io
import os
import sys

import pytest

from graphaligner_amd import aligner, binding, synth
import parity_common as pc
import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg          # independent Python decoder of the GAM framing


def _decode_gam(path):
    out = []
    for m in mg._messages(path):
        d = {}
        for f, v in mg._fields(m):
            d.setdefault(f, []).append(v)
        maps = []
        for f, v in mg._fields(d.get(2, [b""])[0]):
            if f == 2:
                md = dict(mg._fields(v))
                pos = dict(mg._fields(md.get(1, b"")))
                ed = dict(mg._fields(md.get(2, b"")))
                maps.append((int(pos.get(1, 0)), int(pos.get(4, 0)), int(pos.get(2, 0)), int(md.get(5, 0)), int(ed.get(1, 0)), int(ed.get(2, 0)), ed.get(3, b"").decode()))
        out.append(dict(sequence=d[1][0].decode(), name=d[3][0].decode(), score=int(d.get(6, [0])[0]), query_position=int(d.get(7, [0])[0]), mappings=maps))
    return out


def _seed_gam(named_seeds):
    """seed file as the reference reads it: vg.Alignment{name, query_position, path.mapping(0).position{node_id, is_reverse}}"""
    msgs = []
    for name, (node, pos, rev) in named_seeds:
        position = aligner._int_field(1, node) + aligner._int_field(4, int(rev))
        path = aligner._message(2, aligner._message(1, position))
        msgs.append(aligner._message(2, path) + aligner._bytes_field(3, name.encode()) + aligner._int_field(7, pos))
    return aligner.gam_group(msgs)


def _run_driver(tmp_path, lib, ramp=0):
    g = synth.bubble_graph(30000, node_len=32, seed=21)
    reads, seeds = synth.simulate_reads(g, 6, 1200, seed=77, mid_seed=True)
    reads.append(reads[0][:100])                     # too short: the engine asserts (samplingFrequency > 1)
    seeds.append(seeds[0])
    names = ["r%d/x:%d" % (i, i) for i in range(len(reads))] + ["orphan"]
    reads.append("ACGT" * 80)                        # a read without seed hits
    (tmp_path / "g.gfa").write_text(g.gfa())
    with open(tmp_path / "reads.fastq", "w") as f:
        for n, r in zip(names, reads):
            f.write("@%s\n%s\n+\n%s\n" % (n, r, "I" * len(r)))
    (tmp_path / "seeds.gam").write_bytes(_seed_gam([(n, s) for n, s in zip(names, seeds)]))
    p = aligner.parse_args(["-g", str(tmp_path / "g.gfa"), "-f", str(tmp_path / "reads.fastq"), "-s", str(tmp_path / "seeds.gam"),
                            "-a", str(tmp_path / "out.gam"), "-t", "1", "-b", "35"] + (["-B", str(ramp)] if ramp else []))
    p.outputDir = str(tmp_path)
    out, err = io.StringIO(), io.StringIO()
    written = aligner.align_reads(p, lib_path=lib, out=out, err=err)
    return g, names, reads, seeds, written, out.getvalue(), err.getvalue()


def _check_driver(tmp_path, lib):
    g, names, reads, seeds, written, out, err = _run_driver(tmp_path, lib)
    # reads are taken from the back of the list (Aligner.cpp:113-117)
    assert [n for n, _ in written] == [n for n in names[:6]][::-1]
    assert "read orphan has no seed hits" in out and "read orphan has no seed hits" in err
    assert "read %salignment failed (assertion!)" % names[6] in err
    assert "final result has 6 alignments" in err
    got = _decode_gam(str(tmp_path / "out.gam"))
    assert [a["name"] for a in got] == [n for n, _ in written]
    og = ob.OracleGraph(g.nodes, g.edges)
    for a in got:
        i = names.index(a["name"])
        o = og.align(reads[i], [seeds[i]], 35)
        assert a["sequence"] == reads[i] and a["score"] == o["score"] and a["query_position"] == o["query_position"]
        exp = [(m[0] // 2, int(m[1]), m[2], m[3], m[4], m[5], m[6]) for m in o["mappings"]]      # ids halved (Aligner.cpp:83-91)
        assert a["mappings"] == exp
        # per-read files (Aligner.cpp:177-201)
        one = _decode_gam(str(tmp_path / aligner._safe("alignment_0_%s.gam" % a["name"])))
        assert one == [a]
        lines = open(tmp_path / aligner._safe("trace_0_%s.trace" % a["name"])).read().splitlines()
        assert len(lines) == o["trace"].shape[0]
        t0 = o["trace"][0]
        assert lines[0] == "%d %d %d %d %d %s %s" % (t0[0], t0[1], t0[2], t0[3], t0[4], chr(int(t0[5])), chr(int(t0[6])))
    # the library's own GAM encoder writes the same bytes after decompression
    gg = binding.Graph(gfa=g.gfa(), lib_path=lib)
    order = [names.index(n) for n, _ in written]
    b = gg.prepare([reads[i] for i in order], [[seeds[i]] for i in order], 35, names=[names[i] for i in order])
    b.run()
    assert gzip.decompress(b.collect_gam()) == gzip.decompress(open(tmp_path / "out.gam", "rb").read())


def test_driver_on_files_emulated(tmp_path):
    _check_driver(tmp_path, pc.emul_lib_path())


def test_flag_validation():
    err = io.StringIO()
    for argv, msg in ((["-g", "x.gfa", "-f", "r.fq", "-s", "s.gam", "-t", "1", "-b", "1"], "bandwidth must be >= 2"),
                      (["-g", "x.gfa", "-f", "r.fq", "-s", "s.gam", "-t", "0", "-b", "35"], "number of threads must be >= 1"),
                      (["-g", "x.gfa", "-f", "r.fq", "-s", "s.gam", "-t", "1", "-b", "35", "-B", "20"], "backup bandwidth must be higher than initial bandwidth"),
                      (["-g", "x.gfa", "-f", "r.fq", "-t", "1", "-b", "35"], "either initial full band or seed file must be set"),
                      (["-g", "x.gfa", "-f", "r.fq", "-s", "s.gam", "-t", "1", "-b", "35", "-d", "65"], "dynamic row start has to be a multiple of 64")):
        with pytest.raises(SystemExit):
            aligner.parse_args(argv, err=err)
        assert msg in err.getvalue()


def test_reference_smallexample_through_the_driver(tmp_path):
    """the reference's own test files (test/smallexample): its read is 161 bp, so the engine asserts
    (samplingFrequency > 1, GraphAligner.h:906) and the driver reports the read as failed by assertion"""
    import json, shutil
    d = json.load(open(os.path.join(GOLDEN, "ref_smallexample.json")))
    shutil.copy(os.path.join(GOLDEN, "ref_smallexample_sub_test.vg"), tmp_path / "sub_test.vg")
    shutil.copy(os.path.join(GOLDEN, "ref_smallexample_seedalignment.gam"), tmp_path / "seedalignment.gam")
    (tmp_path / "read.fastq").write_text("@%s\n%s\n+\n%s\n" % (d["read_name"], d["read"], "I" * len(d["read"])))
    p = aligner.parse_args(["-g", str(tmp_path / "sub_test.vg"), "-f", str(tmp_path / "read.fastq"), "-s", str(tmp_path / "seedalignment.gam"),
                            "-a", str(tmp_path / "out.gam"), "-t", "1", "-b", "35"])
    p.outputDir = str(tmp_path)
    out, err = io.StringIO(), io.StringIO()
    written = aligner.align_reads(p, lib_path=pc.emul_lib_path(), out=out, err=err)
    assert written == []
    assert "alignment failed (assertion!)" in err.getvalue()
    assert _decode_gam(str(tmp_path / "out.gam")) == []


@pytest.mark.gpu
def test_driver_on_files_gpu(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test needs a GPU")
    _check_driver(tmp_path, None)
