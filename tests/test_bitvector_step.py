"""tools/bitvector_column_step.h (the per-lane column step kept for the lanes = reads kernel) against
the plain cell recurrence on random columns, vertical re-entry included."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bitvector_column_step_matches_the_cell_recurrence():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "bitvector_step_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "emul", "bitvector_step_check.cpp")])
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "agree with the cell recurrence" in res.stdout
