"""Drop-in for the reference's DQNAgent and ReplayBuffer (src/agent.py:17-198).

Same constructor arguments (args.prediction_depth, lr, buffer_size, batch_size, update_every, gamma,
tau), methods and attributes.  Forward, loss, backward, Adam and the soft update are HIP kernels over flat
fp32 parameter buffers; with more than one process the gradient is averaged over ranks (one all-reduce
of the flat buffer) before the optimizer step.
"""
import copy
import os
import random

import numpy as np
import torch

from . import parallel
from ._lib import call, ptr, query
from .model import Dueling_DQN
from .replay import DeviceReplay

device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


class FlatAdam:
    """torch.optim.Adam (default betas/eps, no weight decay) over the network's flat parameter buffer.

    The moments m and v are flat buffers like the parameters.  While every step since they were zero has been a learn
    step over the used rails, the moments of the first dense layer's weights under the empty rows are equal along the
    rows; the optimizer then keeps its data of the two weight blocks in the compact form of
    flm_dqn_adam_soft_update_rails (one pair of moments per (output unit, filter) for the empty rows, packed moments and
    a packed gradient for the used ones: two thirds of the memory traffic, the same bits) and leaves the large buffers
    stale there.  Reading .m or .v brings them up to date and ends that form for good: whoever holds them may write."""

    def __init__(self, network, lr, betas=(0.9, 0.999), eps=1e-8):
        self.network = network
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self._m = self._v = self.grads = None
        self._pairs = None                  # two buffers [2*512*64, 2] (m, v) under the empty rows; [0] is current
        self._packed = None                 # (m, v) under the used rails, [2*512*64*3*rails4] each
        self._compact = None                # (Rp, T1, NA, rails4) while the large buffers are stale in the weight blocks
        self._uniform_from = 0              # moments known equal along the rows from this rail on (None: not known)

    def _ensure(self):
        flat = self.network.flat_params()
        if self._m is None or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self.grads = torch.zeros_like(flat)
            self._compact, self._uniform_from, self._pairs, self._packed = None, 0, None, None
        return flat

    def _moments(self, key, fill):
        call("flm_dqn_adam_moments", ptr(self._m), ptr(self._v), self._m.numel(), *key, ptr(self._pairs[0]),
             ptr(self._packed[0]), ptr(self._packed[1]), fill, torch.cuda.current_stream(self._m.device).cuda_stream)

    def _spread(self):
        """The large moment buffers up to date again."""
        if self._compact is not None:
            self._moments(self._compact, 1)
            self._compact = None

    @property
    def m(self):
        self._spread()
        self._uniform_from = None
        return self._m

    @property
    def v(self):
        self._spread()
        self._uniform_from = None
        return self._v

    def zero_grad(self):
        self._ensure()
        self.grads.zero_()

    def compact_for(self, Rp, T1, NA, rails4):
        """Whether the next step(rails=...) with these sizes will run in the compact form: the learn step then has to
        leave the two weight blocks' gradient packed (fill_padding = 2)."""
        self._ensure()
        return self._uniform_from is not None and self._uniform_from <= int(rails4)

    def step(self, target=None, tau=None, rails=None):
        """One Adam step; with target (the target network's flat parameters) and tau also its soft update with
        the new parameters, in the same pass.  rails = (Rp, T1, NA, rails4, rcb, packed): the learn step ran over the
        used rails and left the first dense layer's gradient under the empty rows to this pass
        (flm_dqn_adam_soft_update_rails); packed: it left the rest of that gradient in the packed form, as it must
        exactly when compact_for(...) said so."""
        flat = self._ensure()
        self.step_count += 1
        st = torch.cuda.current_stream(flat.device).cuda_stream
        hyper = (float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_count)
        if target is not None and rails is not None:
            Rp, T1, NA, rails4, rcb, packed = rails
            key = (int(Rp), int(T1), int(NA), int(rails4))
            if bool(packed) != self.compact_for(*key):
                raise ValueError("the gradient's form does not match the optimizer's (ask compact_for before the learn step)")
            if self._compact is not None and self._compact != key:
                self._spread()
            if packed and self._compact is None:
                dev = flat.device
                if self._pairs is None:
                    self._pairs = [torch.empty((2 * 512 * 64, 2), dtype=torch.float32, device=dev) for _ in range(2)]
                n_packed = 2 * 512 * 64 * 3 * key[3]
                if self._packed is None or self._packed[0].numel() != n_packed:
                    self._packed = None
                    self._packed = tuple(torch.empty((n_packed,), dtype=torch.float32, device=dev) for _ in range(2))
                self._moments(key, 0)
                self._compact = key
            if self._uniform_from is not None:
                self._uniform_from = max(self._uniform_from, key[3])
            extra = (ptr(self._pairs[0]), ptr(self._pairs[1]), ptr(self._packed[0]), ptr(self._packed[1])) if packed \
                else (None, None, None, None)
            call("flm_dqn_adam_soft_update_rails", ptr(flat), ptr(self.grads), ptr(self._m), ptr(self._v), ptr(target),
                 flat.numel(), *hyper, float(tau), *key, ptr(rcb), *extra, st)
            if packed:
                self._pairs.reverse()
            return
        # any other step gives the weights under the empty rows gradients of their own
        self._spread()
        self._uniform_from = None
        if target is None:
            call("flm_dqn_adam", ptr(flat), ptr(self.grads), ptr(self._m), ptr(self._v), flat.numel(), *hyper, st)
        else:
            call("flm_dqn_adam_soft_update", ptr(flat), ptr(self.grads), ptr(self._m), ptr(self._v), ptr(target),
                 flat.numel(), *hyper, float(tau), st)


class DQNAgent:
    """Interacts with and learns from the environment."""

    def __init__(self, args, action_space, bitmap_height, double_dqn=True):
        self.args = args
        self.width = args.prediction_depth + 1
        self.height = bitmap_height
        self.action_space = action_space
        self.double_dqn = double_dqn
        self.qnetwork_local = Dueling_DQN(self.width, self.height, action_space).to(device)
        self.qnetwork_target = copy.deepcopy(self.qnetwork_local)
        self.optimizer = FlatAdam(self.qnetwork_local, lr=self.args.lr)
        self.memory = ReplayBuffer(action_space, self.args.buffer_size, self.args.batch_size, self.width, self.height)
        self.t_step = 0
        self.last_loss = None
        self._ws = None

    def save(self, filename):
        torch.save(self.qnetwork_local.state_dict(), filename + ".local")
        torch.save(self.qnetwork_target.state_dict(), filename + ".target")

    def load(self, filename):
        if os.path.exists(filename + ".local"):
            self.qnetwork_local.load_state_dict(torch.load(filename + ".local", weights_only=True))
        if os.path.exists(filename + ".target"):
            self.qnetwork_target.load_state_dict(torch.load(filename + ".target", weights_only=True))

    def step(self, state, action, reward, next_state, done, train=True):
        self.memory.add(state, action, reward, next_state, done)
        self.t_step = (self.t_step + 1) % self.args.update_every
        if self.t_step == 0:
            if len(self.memory) > self.args.batch_size:
                experiences = self.memory.sample()
                if train:
                    self.learn(experiences, self.args.gamma)

    def act(self, state, eps=0.):
        """Q-values [action_space] (device tensor) of one observation [3*max_rails, depth+1]."""
        state = torch.from_numpy(np.ascontiguousarray(state)).float().unsqueeze(0).unsqueeze(0).to(device)
        self.qnetwork_local.eval()
        with torch.no_grad():
            action_values = self.qnetwork_local(state)
        self.qnetwork_local.train()
        return action_values[0]

    def learn(self, experiences, gamma):
        states, actions, rewards, next_states, dones = experiences
        B = states.size(0)
        local = self.qnetwork_local.flat_params()
        target = self.qnetwork_target.flat_params()
        dev = local.device
        n_ws = query("flm_dqn_learn_workspace", B, self.height, self.width, self.action_space)
        if self._ws is None or self._ws.numel() < n_ws or self._ws.device != dev:
            self._ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        s = states.to(dev, torch.float32).contiguous().view(B, self.height, self.width)
        sn = next_states.to(dev, torch.float32).contiguous().view(B, self.height, self.width)
        a = actions.to(dev, torch.int64).contiguous()
        r = rewards.to(dev, torch.float32).contiguous()
        d = dones.to(dev, torch.float32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        self.optimizer._ensure()                            # (no zero_grad: the kernels write every gradient element)
        st = torch.cuda.current_stream(dev).cuda_stream
        args = (ptr(local), ptr(target), ptr(s), ptr(sn), ptr(a), ptr(r), ptr(d), float(gamma),
                1 if self.double_dqn else 0, B, self.height, self.width, self.action_space, ptr(self._ws),
                ptr(self.optimizer.grads), ptr(loss))
        self._grads_and_exchange("flm_dqn_learn_grads", args, st)
        self.last_loss = loss
        self._update(target)

    def learn_obs(self, experiences, gamma, plane_stride, rails_used=0):
        """learn() on uint8 observations [B,3,plane_stride] (plane 0 int8, heat planes uint8) as a device replay
        memory stores them: same update, no float copies of the next states.  rails_used > 0: the caller vouches that
        rails rails_used.. of every plane of every observation are empty (the padding up to max_rails); the step
        then computes over the used rails and adds what the empty ones contribute in closed form.  With several ranks
        every rank must pass the same rails_used (the largest of theirs): the ranks' gradients are added element by
        element, and what one rank leaves to the closed form another must not have computed from data."""
        states, actions, rewards, next_states, dones = experiences
        B = states.size(0)
        if self.height % 3:
            raise ValueError("observations have three planes: the network height must be 3*max_rails")
        Rp = self.height // 3
        local = self.qnetwork_local.flat_params()
        target = self.qnetwork_target.flat_params()
        dev = local.device
        n_ws = query("flm_dqn_learn_workspace_u8", B, Rp, self.width, self.action_space)
        if self._ws is None or self._ws.numel() < n_ws or self._ws.device != dev:
            self._ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        s, sn = states.contiguous(), next_states.contiguous()
        if s.dtype != torch.uint8 or sn.dtype != torch.uint8 or s.shape[1:] != (3, plane_stride) or s.shape != sn.shape:
            raise ValueError("uint8 observations [B,3,plane_stride] expected")
        a = actions.to(dev, torch.int64).contiguous()
        r = rewards.to(dev, torch.float32).contiguous()
        d = dones.to(dev, torch.float32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        self.optimizer._ensure()                            # (no zero_grad: the kernels write every gradient element)
        st = torch.cuda.current_stream(dev).cuda_stream
        # over the used rails the gradient under the empty rows is a product the optimizer pass can form itself: it is
        # then neither written nor read (only with this class's own optimizer step and soft update, and while the
        # weight blocks' indices stay below 2^24)
        rails4 = query("flm_dqn_learn_rails", B, Rp, self.width, int(rails_used))
        fused = rails4 > 0 and type(self).soft_update is DQNAgent.soft_update and 512 * 64 * self.height < (1 << 24)
        packed = fused and self.optimizer.compact_for(Rp, self.width, self.action_space, rails4)
        args = (ptr(local), ptr(target), ptr(s), ptr(sn), int(plane_stride), ptr(a), ptr(r), ptr(d),
                float(gamma), 1 if self.double_dqn else 0, B, Rp, self.width, self.action_space, int(rails_used),
                ptr(self._ws), ptr(self.optimizer.grads), ptr(loss), (2 if packed else 0) if fused else 1)
        self._grads_and_exchange("flm_dqn_learn_grads_u8_rails", args, st, phased_entry=True,
                                 packed_cols=64 * 3 * rails4 if packed else 0)
        self.last_loss = loss
        if fused:
            off = query("flm_dqn_learn_rails_rcb_offset", B, Rp, self.width, self.action_space)
            self.optimizer.step(target=target, tau=self.args.tau,
                                rails=(Rp, self.width, self.action_space, rails4, self._ws[off:off + 64], packed))
        else:
            self._update(target)

    def _grads_and_exchange(self, entry, args, stream, phased_entry=False, packed_cols=0):
        """The gradients of one learn step and, with several ranks, their average over the ranks (the reference is a
        single process, src/agent.py:14: one rank is its algorithm).  Data parallel: the step runs in two halves;
        the gradients of the dense layers (98 % of the flat buffer) are final after the first and travel while the
        second computes the convolution's, whose small sum follows; both are waited for before the optimizer.
        phased_entry: the entry point itself takes the phase (0 = the whole step) in front of the stream.
        packed_cols > 0: the two weight blocks of the first dense layer hold [512, packed_cols] gradients at their
        start and nothing behind them (fill_padding = 2): only what is there travels."""
        whole = (lambda: call(entry, *args, 0, stream)) if phased_entry else (lambda: call(entry, *args, stream))
        half = (lambda ph: call(entry, *args, ph, stream)) if phased_entry else (lambda ph: call(entry + "_phase", *args, ph, stream))
        if not parallel.distributed():
            whole()
            return
        grads = self.optimizer.grads
        split = query("flm_dqn_grad_split", self.height, self.width, self.action_space)
        ex = self.exchange = parallel.GradientExchange()
        half(1)
        if packed_cols:
            a1_w, block = split, 512 * 64 * self.height          # flat layout: weights [512, 64*height], then 512 biases
            a1_b = a1_w + block
            v1_w = a1_b + 512
            ex.begin(grads[a1_w:a1_w + 512 * packed_cols])       # advantage stream's packed weights
            ex.begin(grads[a1_b:v1_w + 512 * packed_cols])       # its biases and the value stream's packed weights
            ex.begin(grads[v1_w + block:])                       # the value stream's biases and the output layers
        else:
            ex.begin(grads[split:])
        half(2)
        ex.begin(grads[:split])
        ex.finish()

    def _update(self, target):
        """optimizer.step() and soft_update (src/agent.py:113-118); one pass when soft_update is the class's own."""
        if type(self).soft_update is DQNAgent.soft_update:
            self.optimizer.step(target=target, tau=self.args.tau)
        else:
            self.optimizer.step()
            self.soft_update(self.qnetwork_local, self.qnetwork_target, self.args.tau)

    def soft_update(self, local_model, target_model, tau):
        local, target = local_model.flat_params(), target_model.flat_params()
        call("flm_dqn_soft_update", ptr(target), ptr(local), local.numel(), float(tau),
             torch.cuda.current_stream(local.device).cuda_stream)


class ReplayBuffer:
    """The reference's fixed-size buffer of experience tuples with uniform sampling (src/agent.py:134-198), kept in
    device memory: add() uploads one transition (states as float32 [height*width]), sample() draws batch_size of
    them without replacement and returns the reference's five tensors (states and next states [B,1,height,width]
    float, actions long, rewards and dones float) on the device, as fresh tensors the caller may keep (the
    reference stacks new tensors per call, src/agent.py:163-172), in the sampler's random order as random.sample
    returns them in its own."""

    def __init__(self, action_space, buffer_size, batch_size, width, height):
        self.action_space = action_space
        self.batch_size = batch_size
        self.width = width
        self.height = height
        self.buffer_size = buffer_size
        self._pad = (-(height * width)) % 4                    # rows of whole 16-byte pieces
        self._memory = None                                    # device buffers are allocated by the first add()

    @property
    def memory(self):
        if self._memory is None:
            self._memory = DeviceReplay(self.buffer_size, self.height * self.width + self._pad, device,
                                        seed=random.getrandbits(32), dtype=torch.float32)
        return self._memory

    def _row(self, state):
        x = torch.from_numpy(np.ascontiguousarray(state, dtype=np.float32).reshape(1, -1))
        if self._pad:
            x = torch.nn.functional.pad(x, (0, self._pad))
        return x.to(device)

    def add(self, state, action, reward, next_state, done):
        # the three scalars travel as one small upload
        ard = torch.tensor([float(int(action)), float(reward), 1.0 if done else 0.0], dtype=torch.float64).to(device)
        self.memory.add(self._row(state), ard[0:1].to(torch.int64), ard[1:2].to(torch.float32), self._row(next_state),
                        ard[2:3].to(torch.float32))

    def sample(self):
        # DeviceReplay.sample returns buffers of its own that the next call overwrites: hand out copies
        s, a, r, sn, d = self.memory.sample(self.batch_size)
        n = self.height * self.width
        shape = (self.batch_size, 1, self.height, self.width)
        return s[:, :n].reshape(shape).clone(), a.clone(), r.clone(), sn[:, :n].reshape(shape).clone(), d.clone()

    def __len__(self):
        return 0 if self._memory is None else len(self._memory)
