"""The one known-answer vector the reference itself ships: docs/cell_seq.txt + docs/map.txt -> docs/z-bitmaps-0.txt
(agents 1-3; agent 0's rows are hand-annotated in the reference's file).  Unlike every other golden vector it was
not produced through this repository's stand-in for the flatland package.  Checked here: the CPU checker's
rasteriser (src/rail_observations.py:323-407) followed by get_many's one-step shift (:92-93)."""
import types

import numpy as np

from tests.golden_util import docs_kat


def kat_graph(k):
    return types.SimpleNamespace(node_id=k["node_id"], cell_rail=k["cell_rail"],
                                 edge_info=np.ascontiguousarray(k["edge_info"].reshape(-1)), W=k["W"], R=k["R"])


def test_fixture_shape_and_runs():
    k = docs_kat()
    assert (k["H"], k["W"], k["R"], k["T1"]) == (20, 20, 46, 501)
    bm = k["bitmaps"]
    for a in k["clean_agents"]:
        assert (np.abs(bm[a]).sum(0) <= 1).all()                  # at most one rail per time step
        assert (bm[a][:, 0] == 0).all()                           # column 0 is cleared by the shift
        for r in range(k["R"]):                                   # one direction per rail
            v = bm[a][r][bm[a][r] != 0]
            assert len(set(v.tolist())) <= 1


def test_checker_reproduces_the_reference_dump():
    from oracle import oracle as orc
    k = docs_kat()
    g = kat_graph(k)
    for a in k["clean_agents"]:
        cells = k["cell_seq"][a]
        tr, tc = cells[-1]
        raw, err = orc.bitmap_from_cells(g, cells, tr, tc, k["T1"])
        assert err == "", (a, err)
        got = orc.shift_right(raw.copy())
        assert np.array_equal(got, k["bitmaps"][a]), a
