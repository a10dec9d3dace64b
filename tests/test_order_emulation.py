"""The device program has to reproduce three ORDERS of the reference, because the traceback
starts from the last minimum in processing order (GraphAligner.h:922,931):
  * iteration order of the frozen slice's std::unordered_map (NodeSlice.h:724-740),
  * pop order of std::priority_queue among equal priorities (GraphAligner.h:1115),
  * Tarjan emission order (covered end to end by the parity cases).
Here the first two are checked in isolation: the device code (run through the host emulation,
tests/emul) against the real libstdc++ containers (via the oracle library) and, for the hash
order, against the reference's own NodeSlice where oracle/_ref is present."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import parity_common as pc


@pytest.fixture(scope="module")
def emul():
    return C.CDLL(pc.emul_lib_path())


def test_hash_iteration_order(emul):
    L = ob.lib()
    L.gao_frozen_order.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    rng = np.random.default_rng(11)
    ref = ob.reflib()
    for trial in range(400):
        n = int(rng.integers(1, 257)) if trial % 2 else int(rng.integers(1, 65))
        universe = int(rng.choice([300, 5000, 200000]))
        keys = rng.choice(universe, size=n, replace=False).astype(np.uint32)
        out = np.zeros(n, dtype=np.int32)
        assert emul.ga_emul_hash_order(keys.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)) == n
        k64 = keys.astype(np.int64)
        o = np.zeros(n, dtype=np.int64)
        assert L.gao_frozen_order(ob._p(k64), n, universe, ob._p(o)) == n
        assert (keys[out] == o).all()
        if n <= 64:
            # the lane-parallel form used for bands of up to 64 nodes
            out2 = np.zeros(n, dtype=np.int32)
            assert emul.ga_emul_hash_order_lanes(keys.ctypes.data_as(C.c_void_p), n, out2.ctypes.data_as(C.c_void_p)) == n
            assert (out2 == out).all(), (n, keys, out, out2)
        if ref is not None and trial % 8 == 0:
            o2 = np.zeros(n, dtype=np.int64)
            assert ref.ref_frozen_order(ob._p(k64), n, universe, ob._p(o2)) == n
            assert (keys[out] == o2).all()


def test_priority_queue_tie_order(emul):
    L = ob.lib()
    L.gao_pq_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(12)
    for trial in range(300):
        n_ops = int(rng.integers(1, 400))
        nodes = rng.integers(0, 1000, size=n_ops).astype(np.uint32)
        prios = rng.integers(0, int(rng.choice([3, 10, 100])), size=n_ops).astype(np.int32)
        pops = rng.random(n_ops) < 0.35
        prios[pops] = -1
        a = np.zeros(n_ops, dtype=np.uint32)
        b = np.zeros(n_ops, dtype=np.uint32)
        ka = emul.ga_emul_heap(nodes.ctypes.data_as(C.c_void_p), prios.ctypes.data_as(C.c_void_p), n_ops, a.ctypes.data_as(C.c_void_p))
        kb = L.gao_pq_order(nodes.ctypes.data_as(C.c_void_p), prios.ctypes.data_as(C.c_void_p), n_ops, b.ctypes.data_as(C.c_void_p))
        assert ka == kb
        if ka >= 0 and int((prios >= 0).sum()) <= 256:
            c = np.zeros(n_ops, dtype=np.uint32)
            kc = emul.ga_emul_heap_lanes(nodes.ctypes.data_as(C.c_void_p), prios.ctypes.data_as(C.c_void_p), n_ops, c.ctypes.data_as(C.c_void_p))
            assert kc == ka and (c[:kc] == a[:ka]).all()
        assert (a[:ka] == b[:kb]).all()
