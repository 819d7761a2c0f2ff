"""Drop-in for the reference's Dueling_DQN (src/model.py:11-43): same constructor, same state_dict keys
and shapes, same default initialisation; forward runs the hand-written matrix-core kernels (fp32 values, no autograd)."""
import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr, query


class Dueling_DQN(nn.Module):
    def __init__(self, width, height, action_space):
        super(Dueling_DQN, self).__init__()
        action_space = int(action_space)
        if not 1 <= action_space <= _lib.DQN_MAX_ACTIONS:
            raise ValueError("action_space must be 1..%d for the device kernels" % _lib.DQN_MAX_ACTIONS)
        self.action_space = action_space
        self.width, self.height = int(width), int(height)
        # same submodules in the same order as the reference => identical default init and state_dict keys
        self.conv1 = nn.Conv2d(in_channels=1, out_channels=64, kernel_size=(1, width))
        self.fc1_adv = nn.Linear(in_features=64 * height, out_features=512)
        self.fc1_val = nn.Linear(in_features=64 * height, out_features=512)
        self.fc2_adv = nn.Linear(in_features=512, out_features=action_space)
        self.fc2_val = nn.Linear(in_features=512, out_features=1)
        self.relu = nn.ReLU()
        self._flat = None
        self._ws = {}

    # ---- flat parameter buffer -------------------------------------------------------------------------
    def flat_params(self):
        """One contiguous fp32 device buffer; the module's parameters are views into it."""
        # (self.parameters() in registration order, without the module walk: this runs several times per learn step)
        params = [self.conv1.weight, self.conv1.bias, self.fc1_adv.weight, self.fc1_adv.bias, self.fc1_val.weight,
                  self.fc1_val.bias, self.fc2_adv.weight, self.fc2_adv.bias, self.fc2_val.weight, self.fc2_val.bias]
        dev = params[0].device
        if dev.type != "cuda":
            raise _lib.FlmError("Dueling_DQN must live on a GPU (call .to('cuda')): there is no CPU forward")
        ok = self._flat is not None and self._flat.device == dev
        if ok:
            off = 0
            base = self._flat.data_ptr()
            for p in params:
                if p.data_ptr() != base + 4 * off:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            n = sum(p.numel() for p in params)
            assert n == query("flm_dqn_param_count", self.height, self.width, self.action_space)
            flat = torch.empty(n, dtype=torch.float32, device=dev)
            off = 0
            for p in params:
                flat[off:off + p.numel()].copy_(p.data.reshape(-1).float())
                p.data = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
            self._flat = flat
        return self._flat

    def __deepcopy__(self, memo):
        new = Dueling_DQN(self.width, self.height, self.action_space)
        new.load_state_dict(self.state_dict())
        dev = next(self.parameters()).device
        new.to(dev)
        new.train(self.training)
        return new

    def _workspace(self, B):
        """Scratch floats for a forward pass of B rows.  The buffer only grows (with some headroom): rollouts
        ask for a different B every step, and handing a buffer of a new size back and forth to the caching
        allocator costs device allocations in the middle of the step."""
        dev = self._flat.device
        ws = self._ws.get("buf")
        if ws is not None and ws.device == dev and B <= self._ws["rows"]:
            return ws
        need = query("flm_dqn_forward_workspace", B, self.height)
        if ws is None or ws.device != dev or ws.numel() < need:
            self._ws = {}
            ws = None                                           # release the old buffer before taking a larger one
            rows = B + B // 4 + 1
            grow = query("flm_dqn_forward_workspace", rows, self.height)
            self._ws = {"buf": torch.empty(grow, dtype=torch.float32, device=dev), "rows": rows}
            ws = self._ws["buf"]
        else:
            self._ws["rows"] = max(self._ws["rows"], B)
        return ws

    def forward(self, x):
        flat = self.flat_params()
        x = x.to(device=flat.device, dtype=torch.float32).contiguous()
        B = x.size(0)
        x = x.view(B, self.height, self.width)
        q = torch.empty((B, self.action_space), dtype=torch.float32, device=flat.device)
        call("flm_dqn_forward", ptr(flat), ptr(x), B, self.height, self.width, self.action_space,
             ptr(self._workspace(B)), ptr(q),
             torch.cuda.current_stream(flat.device).cuda_stream)
        return q

    def forward_obs(self, obs_u8, plane_stride, rails_used=0, out=None, flat=None, stream=None):
        """Q-values [n,action_space] straight from uint8 observations [n,3,plane_stride] (plane 0 int8, heat planes uint8,
        rows of `width` bytes): the convolution reads the bytes, no float copy of the input is made.
        rails_used > 0: the caller vouches that rails rails_used.. of every plane are empty (the padding up to
        max_rails); they are then not computed but enter through the first dense layer's bias.
        out: a float32 buffer of at least n rows for the result; flat: what flat_params() returned a moment ago;
        stream: the raw stream to launch on (a caller in a hurry has these at hand)."""
        if flat is None:
            flat = self.flat_params()
        if self.height % 3:
            raise ValueError("observations have three planes: the network height must be 3*max_rails")
        if not obs_u8.is_contiguous():
            obs_u8 = obs_u8.contiguous()
        n = obs_u8.size(0)
        q = torch.empty((n, self.action_space), dtype=torch.float32, device=flat.device) if out is None else out[:n]
        call("flm_dqn_forward_u8_rails", ptr(flat), ptr(obs_u8), n, self.height // 3, self.width, int(plane_stride),
             self.action_space, int(rails_used), ptr(self._workspace(n)), ptr(q),
             torch.cuda.current_stream(flat.device).cuda_stream if stream is None else stream)
        return q

