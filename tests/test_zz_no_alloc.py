"""flm_bitmaps_static (the entry point that allocates nothing: its column workspace is a fixed array inside the
library) against flm_bitmaps, through the C ABI, on two calls in a row, the second with a smaller input.
Bit-exact.  Needs an MI355X."""
import gc

import numpy as np
import pytest

from tests.golden_util import load_env

pytestmark = pytest.mark.gpu


def _both(b, n_paths, agent_flat, shift):
    import torch
    from flatland_marl_amd._lib import call, ptr
    dev = b.device
    s = b.stream.cuda_stream
    common = (ptr(b.path_cell), ptr(b.path_len), ptr(agent_flat), ptr(b.pos_cell), ptr(b.target_cell), ptr(b.tpc),
              ptr(b.node_id), ptr(b.cell_rail), ptr(b.edge_info), b.E, b.A, n_paths, b.H, b.W, b.capR, b.Rp, b.T1,
              b.plane_stride, b.Lcap, shift)
    want = torch.full((n_paths, b.plane_stride), 7, dtype=torch.int8, device=dev)
    want_err = torch.full((n_paths,), -7, dtype=torch.int32, device=dev)
    ws = torch.empty((n_paths, b.T1), dtype=torch.int16, device=dev)
    call("flm_bitmaps", *common, ptr(ws), ptr(want), ptr(want_err), s)
    got = torch.full((n_paths, b.plane_stride), 9, dtype=torch.int8, device=dev)
    got_err = torch.full((n_paths,), -9, dtype=torch.int32, device=dev)
    # What can see an allocation made inside the library is the driver's own count of free device memory
    # (hipMemGetInfo); torch's counter only sees torch's allocator and moves whenever Python's collector frees
    # tensors of earlier tests, so the collector is run first and held off around the call.
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        free_before = torch.cuda.mem_get_info(dev)[0]
        torch_before = torch.cuda.memory_allocated(dev)
        call("flm_bitmaps_static", *common, ptr(got), ptr(got_err), s)
        torch.cuda.synchronize()
        free_after = torch.cuda.mem_get_info(dev)[0]
        torch_after = torch.cuda.memory_allocated(dev)
    finally:
        gc.enable()
    alloc = (free_before, free_after, torch_before, torch_after)
    return want.cpu().numpy(), want_err.cpu().numpy(), got.cpu().numpy(), got_err.cpu().numpy(), alloc


def test_static_workspace_entry_point_matches_flm_bitmaps():
    import torch
    import flatland_marl_amd as fm
    from flatland_marl_amd import _lib
    fx = load_env("Test_2_Level_0")                   # 80 agents, 70 rails
    b = fm.ObservationBatch(fx["grid"], fx["initial_position"], fx["direction"], fx["target"], fx["speed"], 150)
    b.build_graph()
    b.compute_distances()
    b.shortest_paths()
    # first call: every agent's path
    want, werr, got, gerr, alloc1 = _both(b, b.E * b.A, None, 1)
    assert np.array_equal(werr, gerr) and np.array_equal(want, got)
    assert np.abs(want).sum() > 0
    # second call, smaller: 7 agents' paths, not shifted (the workspace still holds the first call's columns)
    pick = np.array([3, 0, 41, 17, 79, 5, 22], np.int32)
    small = fm.ObservationBatch(fx["grid"], fx["initial_position"][pick], fx["direction"][pick], fx["target"][pick],
                                fx["speed"][pick], 150)
    small.build_graph()
    small.compute_distances()
    small.shortest_paths()
    want2, werr2, got2, gerr2, alloc2 = _both(small, 7, None, 0)
    assert np.array_equal(werr2, gerr2) and np.array_equal(want2, got2)
    # the planes are the ones the checked path produces for those agents (shift 0 vs shift 1: one column apart)
    full = want.reshape(b.A, -1)[:, : b.Rp * b.T1].reshape(b.A, b.Rp, b.T1)
    part = want2.reshape(7, -1)[:, : small.Rp * small.T1].reshape(7, small.Rp, small.T1)
    for i, a in enumerate(pick):
        if werr[a] == 0:
            assert np.array_equal(part[i][:, :-1], full[a][:, 1:]), a
    # a call beyond the fixed workspace is refused on the host, before anything is launched
    lib = _lib.load()
    too_many = _lib.load().flm_bitmaps_static
    n_big = (2 * 1024 * 1024) // b.T1 + 1
    rc = too_many(b.path_cell.data_ptr(), b.path_len.data_ptr(), None, b.pos_cell.data_ptr(), b.target_cell.data_ptr(),
                  b.tpc.data_ptr(), b.node_id.data_ptr(), b.cell_rail.data_ptr(), b.edge_info.data_ptr(), b.E, b.A, n_big,
                  b.H, b.W, b.capR, b.Rp, b.T1, b.plane_stride, b.Lcap, 1, b.path_cell.data_ptr(), b.path_len.data_ptr(),
                  b.stream.cuda_stream)
    assert rc == -3
    # last, so that a quirk of a counter can never hide the comparisons above: neither call took device memory
    for free_before, free_after, torch_before, torch_after in (alloc1, alloc2):
        assert free_after >= free_before, (free_before, free_after)
        assert torch_after <= torch_before, (torch_before, torch_after)
