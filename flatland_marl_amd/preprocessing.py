"""Drop-in for the reference's ObsPreprocessor (src/preprocessing.py:3-74), heat maps on the GPU."""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr


class ObsPreprocessor:
    def __init__(self, max_rails, reorder_rails):
        self.max_rails = max_rails
        self.reorder_rails = reorder_rails
        if not torch.cuda.is_available():
            raise _lib.FlmError("no GPU visible: the heat maps have no CPU implementation")
        _lib.load()
        self.device = torch.device("cuda:0")

    def get_obs(self, handle, bitmap, maps):
        """[3*max_rails, T1] float64: the agent's plane (possibly an alternative one), then the sums of the
        positive and of the negative entries of all other agents' planes."""
        maps = np.asarray(maps)
        bitmap = np.asarray(bitmap)
        A, R, T1 = maps.shape
        Rp = self.max_rails
        if R > Rp or bitmap.shape[0] > Rp:
            raise ValueError("could not broadcast input array from shape (%d,%d) into shape (%d,%d)" % (R, T1, Rp, T1))
        if A > 255:
            raise ValueError("at most 255 agents")
        stride = (Rp * T1 + 15) // 16 * 16
        host = np.zeros((A + 1, stride), np.int8)
        host[:A, :Rp * T1].reshape(A, Rp, T1)[:, :R] = maps
        host[A, :Rp * T1].reshape(Rp, T1)[:bitmap.shape[0]] = bitmap
        dev = torch.as_tensor(host, device=self.device)
        req = torch.tensor([handle], dtype=torch.int32, device=self.device)
        out = torch.empty((1, 3, stride), dtype=torch.uint8, device=self.device)
        call("flm_observations_req", ptr(dev), ptr(req), ptr(dev[A:]), 1, 1, A, stride, ptr(out),
             torch.cuda.current_stream(self.device).cuda_stream)
        if self.reorder_rails:
            # src/preprocessing.py:32-59; as there, the caller's bitmap ends up reordered too
            res = torch.empty_like(out)
            err = torch.zeros((1,), dtype=torch.int32, device=self.device)
            call("flm_reorder_rails", ptr(out), ptr(res), 1, Rp, T1, stride, ptr(err),
                 torch.cuda.current_stream(self.device).cuda_stream)
            if int(err.item()):
                raise RuntimeError("reorder_rails: agent %d stays on one rail up to the last predicted step; "
                                   "the reference's loop (src/preprocessing.py:49-52) never ends here" % handle)
            out = res
        o = out[0, :, :Rp * T1].cpu().numpy()
        if self.reorder_rails and bitmap.flags.writeable:
            bitmap[...] = o[0].view(np.int8).reshape(Rp, T1)[:bitmap.shape[0]]
        state = o.astype(np.float64)
        state[0] = o[0].view(np.int8)
        return state.reshape(3 * Rp, T1)
