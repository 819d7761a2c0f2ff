"""The reference's own known-answer vector (docs/cell_seq.txt + docs/map.txt -> docs/z-bitmaps-0.txt, agents 1-3)
through the HIP rasteriser: flm_columns (column form) and flm_bitmaps (dense planes), called through the C ABI
with the cell -> rail / switch tables of the reference's map dump.  Needs an MI355X."""
import numpy as np
import pytest

from tests.golden_util import docs_kat

pytestmark = pytest.mark.gpu


def test_hip_rasteriser_reproduces_the_reference_dump():
    import torch
    from flatland_marl_amd._lib import call, ptr
    k = docs_kat()
    H, W, R, T1 = k["H"], k["W"], k["R"], k["T1"]
    agents = [int(a) for a in k["clean_agents"]]
    n = len(agents)
    dev = torch.device("cuda", 0)
    i32 = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.int32), device=dev)
    # one path element per time step (one time step per cell): the predicted cell sequence is the path itself
    seq = k["cell_seq"][agents]
    cells = (seq[..., 0] * W + seq[..., 1]).astype(np.int32)                     # [n,T1]
    Lcap = T1 + 3
    path_cell = np.zeros((n, Lcap), np.int32)
    path_cell[:, :T1] = cells
    args = dict(path_cell=i32(path_cell), path_len=i32(np.full(n, T1)), pos=i32(cells[:, 0]), tgt=i32(cells[:, -1]),
                tpc=i32(np.ones(n)), node_id=i32(k["node_id"]), cell_rail=i32(k["cell_rail"]), edge=i32(k["edge_info"]))
    cols = torch.empty((n, T1), dtype=torch.int16, device=dev)
    err = torch.empty((n,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    # E = 1 environment with A = n agents, n_paths = n, capR = R, Rp = R, shift = 1 (get_many, :92-93)
    call("flm_columns", ptr(args["path_cell"]), ptr(args["path_len"]), None, ptr(args["pos"]), ptr(args["tgt"]),
         ptr(args["tpc"]), ptr(args["node_id"]), ptr(args["cell_rail"]), ptr(args["edge"]), 1, n, n, H, W, R, R, T1, Lcap, 1,
         ptr(cols), ptr(err), s)
    stride = (R * T1 + 255) // 256 * 256
    planes = torch.empty((n, stride), dtype=torch.int8, device=dev)
    cols2 = torch.empty_like(cols)
    err2 = torch.empty_like(err)
    call("flm_bitmaps", ptr(args["path_cell"]), ptr(args["path_len"]), None, ptr(args["pos"]), ptr(args["tgt"]),
         ptr(args["tpc"]), ptr(args["node_id"]), ptr(args["cell_rail"]), ptr(args["edge"]), 1, n, n, H, W, R, R, T1, stride,
         Lcap, 1, ptr(cols2), ptr(planes), ptr(err2), s)
    torch.cuda.synchronize()
    assert not err.cpu().numpy().any() and not err2.cpu().numpy().any()
    got_cols = cols.cpu().numpy()
    assert np.array_equal(got_cols, cols2.cpu().numpy())
    got_planes = planes[:, : R * T1].reshape(n, R, T1).cpu().numpy()
    for i, a in enumerate(agents):
        want = k["bitmaps"][a]
        assert np.array_equal(got_planes[i], want), a
        # the column form says the same: rail*2 + (sign > 0) where a column is occupied, -1 elsewhere
        occ = np.abs(want).sum(0) > 0
        rail = np.abs(want).argmax(0)
        sign = want[rail, np.arange(T1)]
        assert np.array_equal(got_cols[i], np.where(occ, rail * 2 + (sign > 0), -1)), a
