"""Device-resident data and optimiser state of the IRL reward learning (reference ac_irl.py:804-954).

The reference keeps the demonstrations and D_samp as Python ``list[n] of list[15] of (pi [d], P [d,d])`` and rebuilds a
feed_dict from them for every ``update_reward`` (ac_irl.py:814-840).  Here both live on the GPU as two fp32 tensors
(states ``[rows, 15, d]``, actions ``[rows, 15, d, d]``); a training batch is a list of store ROWS handed to
``mfg_reward_net_train_step`` by value, and the Python list view exists only when a caller reads it.

  * :class:`TrajectoryStore` -- the store with the FIFO order of ``outerloop`` (ac_irl.py:927-932);
  * :class:`RewardTrainer` -- flat parameter / Adam buffers of a ``networks.RewardNet`` (the module's parameters become
    views into the flat buffer, so the forward kernel, ``state_dict`` and the trainer see the same memory) and the
    two-launch HIP training step.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


class TrajectoryStore:
    """Trajectories of `steps` (state, action) pairs on the device, in a logical (list) order.

    ``rows[i]`` is the physical row of logical trajectory i; dropping the oldest trajectories frees their rows for the
    next ``push`` (ring behaviour without moving data).  ``to_list()`` gives the reference's list-of-lists view (float64
    NumPy pairs), cached until the store changes."""

    def __init__(self, d, steps, device, capacity=0):
        self.d, self.steps, self.device = int(d), int(steps), torch.device(device)
        self.state = torch.empty(0, self.steps, self.d, dtype=torch.float32, device=self.device)
        self.action = torch.empty(0, self.steps, self.d, self.d, dtype=torch.float32, device=self.device)
        self.rows = []
        self._free = []
        self._list = None
        self.version = 0
        if capacity:
            self._grow(capacity)

    def __len__(self):
        return len(self.rows)

    def _grow(self, need):
        cap = self.state.shape[0]
        if cap >= need:
            return
        new = max(need, 2 * cap, 8)
        s = torch.empty(new, self.steps, self.d, dtype=torch.float32, device=self.device)
        a = torch.empty(new, self.steps, self.d, self.d, dtype=torch.float32, device=self.device)
        if cap:
            s[:cap].copy_(self.state)
            a[:cap].copy_(self.action)
        self.state, self.action = s, a
        self._free.extend(range(cap, new))

    def _changed(self):
        self._list = None
        self.version += 1

    def clear(self):
        self._free = list(range(self.state.shape[0]))
        self.rows = []
        self._changed()

    def push(self, states, actions, drop=0):
        """Append n trajectories (device or host tensors [n, steps(+1), d], [n, steps, d, d]) after dropping the `drop`
        oldest ones: D_samp <- (D_samp + D_traj)[drop:]  (ac_irl.py:927-932)."""
        n = int(states.shape[0])
        if actions.shape != (n, self.steps, self.d, self.d) or states.shape[1] < self.steps or states.shape[2] != self.d:
            raise ValueError('TrajectoryStore.push: bad shapes %s / %s' % (tuple(states.shape), tuple(actions.shape)))
        total = self.rows + [None] * n
        dropped, kept = total[:drop], total[drop:]
        self._free.extend(r for r in dropped if r is not None)
        n_new = sum(1 for r in kept if r is None)
        if n_new > len(self._free):
            self._grow(self.state.shape[0] + n_new - len(self._free))
        first_new = n - n_new                      # new trajectories that were dropped again never land
        new_rows = [self._free.pop(0) for _ in range(n_new)]
        if n_new:
            idx = torch.as_tensor(new_rows, dtype=torch.int64, device=self.device)
            self.state.index_copy_(0, idx, states[first_new:, :self.steps].to(self.device, torch.float32))
            self.action.index_copy_(0, idx, actions[first_new:].to(self.device, torch.float32))
        it = iter(new_rows)
        self.rows = [r if r is not None else next(it) for r in kept]
        self._changed()

    def assign_list(self, trajs):
        """Replace the content by a reference-style list of trajectories (each a list of (state, action) pairs)."""
        self.clear()
        n = len(trajs)
        if n == 0:
            return
        for tr in trajs:
            if len(tr) != self.steps:
                raise ValueError('TrajectoryStore: every trajectory needs %d (state, action) pairs, got %d' % (self.steps, len(tr)))
        s = np.array([[np.asarray(p[0], dtype=np.float32) for p in tr] for tr in trajs], dtype=np.float32)
        a = np.array([[np.asarray(p[1], dtype=np.float32) for p in tr] for tr in trajs], dtype=np.float32)
        self.push(torch.from_numpy(s), torch.from_numpy(a))

    def gather(self, logical=None):
        """(states [n, steps, d], actions [n, steps, d, d]) of the given logical trajectories (default: all), in order."""
        rows = self.rows if logical is None else [self.rows[i] for i in logical]
        idx = torch.as_tensor(rows, dtype=torch.int64, device=self.device)
        return self.state.index_select(0, idx), self.action.index_select(0, idx)

    def gather_flat(self):
        """(states [n*steps, d], actions [n*steps, d, d]) of ALL trajectories in logical order, cached until the store changes
        (reward_iteration evaluates the whole store every `iter_check` updates: D_samp changes once per outer iteration, the
        demonstrations never)."""
        if getattr(self, '_flat_version', None) != self.version:
            st, ac = self.gather()
            self._flat = (st.reshape(-1, self.d), ac.reshape(-1, self.d, self.d))
            self._flat_version = self.version
        return self._flat

    def to_list(self):
        if self._list is None:
            if not self.rows:
                self._list = []
            else:
                s, a = self.gather()
                s = s.cpu().numpy().astype(np.float64)
                a = a.cpu().numpy().astype(np.float64)
                self._list = [[(s[m, t], a[m, t]) for t in range(self.steps)] for m in range(s.shape[0])]
        return self._list


class RewardTrainer:
    """Flat fp32 parameter / Adam-moment buffers of a RewardNet and the HIP training step (tf.train.AdamOptimizer
    semantics, ac_irl.py:417: beta1 0.9, beta2 0.999, epsilon 1e-8)."""

    BETA1, BETA2, EPS = 0.9, 0.999, 1e-8

    def __init__(self, net, lr):
        self.net = net
        self.lr = float(lr)
        params = list(net.parameters())
        self.device = params[0].device
        self.dims = (net.d, net.conv1.kernel_size[0], net.conv2.out_channels, net.conv2.kernel_size[0], net.fc3.out_features,
                     net.fc4.out_features)
        self.flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
        off = 0
        for p in params:                                      # module parameters become views into the flat buffer
            n = p.numel()
            p.data = self.flat[off:off + n].view(p.shape)
            off += n
        offs = (C.c_int64 * 11)()
        L.check(L.lib().mfg_reward_net_param_offsets(*self.dims, offs), 'mfg_reward_net_param_offsets')
        if int(offs[10]) != self.flat.numel():
            raise L.MfgError('RewardTrainer: the module has %d parameters, the kernel layout %d' % (self.flat.numel(), int(offs[10])))
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.grad = torch.zeros_like(self.flat)
        self.stats = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.step_count = 0
        self._ws = None

    def _workspace(self, n_transitions):
        need = L.lib().mfg_reward_net_train_workspace_bytes(*self.dims, int(n_transitions))
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.device)
        return self._ws

    def step(self, demo, demo_rows, gen, gen_rows, demo_divisor, seed, grad_only=False):
        """One update_reward on stores `demo` / `gen` (TrajectoryStore) with the PHYSICAL rows of the sampled trajectories.
        grad_only: leave the gradient in self.grad (the caller all-reduces it and calls apply_grad)."""
        from .ops import _stream
        net = self.net
        nd, ng = len(demo_rows), len(gen_rows)
        ws = self._workspace((nd + ng) * demo.steps)
        keep = float(net.keep_prob) if net.use_dropout else 1.0
        dr = (C.c_int32 * max(nd, 1))(*demo_rows)
        gr = (C.c_int32 * max(ng, 1))(*gen_rows)
        if not grad_only:
            self.step_count += 1
        L.check(L.lib().mfg_reward_net_train_step(
            self.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), *self.dims,
            demo.state.data_ptr(), demo.action.data_ptr(), dr, nd, gen.state.data_ptr(), gen.action.data_ptr(), gr, ng,
            int(demo.steps), int(demo_divisor), keep, 1 if net.use_l1l2 else 0, int(seed) & 0xFFFFFFFFFFFFFFFF, self.lr,
            self.BETA1, self.BETA2, self.EPS, max(self.step_count, 1), 1 if grad_only else 0,
            self.grad.data_ptr() if grad_only else None, self.stats.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream()),
            'mfg_reward_net_train_step')

    def apply_grad(self):
        from .ops import _stream
        self.step_count += 1
        L.check(L.lib().mfg_reward_net_adam(self.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.grad.data_ptr(),
                                            self.flat.numel(), self.lr, self.BETA1, self.BETA2, self.EPS, self.step_count,
                                            _stream()), 'mfg_reward_net_adam')

    def state_dict(self):
        return {'m': self.m.detach().cpu().clone(), 'v': self.v.detach().cpu().clone(), 'step': int(self.step_count),
                'stats': self.stats.detach().cpu().clone()}        # (loss / first / second term of the last update)

    def load_state_dict(self, st):
        self.m.copy_(st['m'].to(self.device))
        self.v.copy_(st['v'].to(self.device))
        self.step_count = int(st['step'])
        if st.get('stats') is not None:
            self.stats.copy_(st['stats'].to(self.device))

    def seed_from_torch_adam(self, optimizer, params):
        """Moments and step count from a torch.optim.Adam state over `params` (the module's parameters, in the flat buffer's
        order): checkpoints written before the HIP training step existed carry only that.  Parameters the optimiser never
        stepped keep zero moments; warns if there is nothing to take."""
        import warnings
        off, steps = 0, 0
        self.m.zero_(); self.v.zero_()
        for p in params:
            n = p.numel()
            st = optimizer.state.get(p)
            if st:
                self.m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                self.v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                steps = max(steps, int(st['step']))
            off += n
        self.step_count = steps
        if steps == 0:
            warnings.warn('checkpoint carries no reward-trainer state and an empty optimiser state: the resumed reward '
                          'learning starts with zero Adam moments', RuntimeWarning, stacklevel=3)
