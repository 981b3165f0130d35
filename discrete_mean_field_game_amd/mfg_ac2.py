"""Drop-in ``actor_critic`` with the call surface of the reference's ``mfg_ac2.actor_critic``
(mfg_ac2.py:23), running the hot path in HIP kernels over a batch of independent trajectories.

Same constructor / method names, keyword names and defaults as the reference:
``actor_critic(theta, shift, alpha_scale, d)``, ``sample_action(pi)``, ``calc_reward(P, pi, d)``,
``calc_features``, ``calc_value``, ``calc_gradient_vectorized``, ``train(...)``, ``generate_trajectory``,
``JSD``, ``evaluate``, ``gridsearch``, plus the public attributes ``theta, shift, alpha_scale, d, w,
mat_pi0, num_start_samples, mat_alpha, mat_alpha_deriv``.  Extensions are keyword-only:

  pi0 / path_to_dir : start-state table (array, or a directory of trend_distribution_day%d.csv files;
                      default cwd/train_normalized_round2 like mfg_ac2.py:39, synthetic if absent)
  batch             : trajectories stepped in lock-step per episode (reference = 1)
  rng               : 'philox'  in-kernel counter-based Dirichlet sampler (production path)
                      'numpy'   gamma variates drawn on the host from the process-global legacy
                                np.random stream in the reference's exact order (mfg_ac2.py:242, :466),
                                so a seeded batch-1 run retraces the reference; the math still runs on
                                the GPU
  precision         : 'mixed' (fp32 hardware transcendentals, fp64 sums; default) or 'f64' (strict)
  episode_steps     : env steps per episode (reference: 15, mfg_ac2.py:478)
  update_every      : 'step' (reference semantics: theta, w move after every env step; batch-mean
                      gradient) or 'rollout' (one fused T-step kernel + one update per episode)
  group             : torch.distributed process group; the batch is the GLOBAL batch and is sharded
                      across ranks, gradients are summed with one all-reduce per update.

There is no CPU fallback: every numeric method calls the C ABI in include/mfg_hip.h.
"""
from __future__ import annotations

import os
import warnings

import numpy as np
import torch

from . import _lib as L
from . import ops
from .parallel import (all_reduce_gradients_, broadcast_start_indices, current_shard, forget_native_comm, lr_scales,
                       native_comm)

EPISODE_STEPS = 15  # mfg_ac2.py:478


def _as_np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _with_ctx(method):
    """Public methods that reach the HIP library run with the instance's own context bound (ops.Context: its status word),
    so that two instances on one device cannot stop each other (include/mfg_hip.h, mfg_ctx_bind).  The binding the calling
    thread had before is restored on the way out: free-function ops.* calls and other instances' unbound reads made afterwards
    do not report into whichever instance happened to run last."""
    import functools

    @functools.wraps(method)
    def bound(self, *args, **kwargs):
        prev = self._ctx.bind_scoped()
        try:
            return method(self, *args, **kwargs)
        finally:
            self._ctx.restore(prev)
    return bound


class actor_critic:

    def __init__(self, theta=8.86349, shift=0.16, alpha_scale=12000, d=21, *, pi0=None, path_to_dir=None,
                 batch=1, rng='philox', seed=0, update_every='step', reward='mfg_ac2', precision='mixed', device=None,
                 group=None, verbose=1, check_finite=False, episode_steps=EPISODE_STEPS):
        if rng not in ('philox', 'numpy'):
            raise ValueError("rng must be 'philox' or 'numpy'")
        if update_every not in ('step', 'rollout'):
            raise ValueError("update_every must be 'step' or 'rollout'")
        if not torch.cuda.is_available():
            raise L.MfgError('actor_critic needs a ROCm GPU: the HIP hot path has no CPU fallback')
        L.lib()
        ops.init()
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self._ctx = ops.Context(self.device)             # this instance's library state (status word), bound by its methods
        self._ctx_prev = self._ctx.bind_scoped()         # ... and for the rest of the constructor (_end_init restores)
        self.shift = shift
        self.alpha_scale = alpha_scale
        self.d = d
        self.rng = rng
        self.seed = int(seed)
        self.update_every = update_every
        self.reward_kind = {'mfg_ac2': L.REWARD_MFG_AC2, 'synthetic': L.REWARD_SYNTHETIC}[reward]
        if precision not in ('mixed', 'f64'):
            raise ValueError("precision must be 'mixed' or 'f64'")
        self.precision = precision
        self.group = group
        self.verbose = verbose
        # numeric sanitiser: the reference turns every NumPy FP warning into an exception (mfg_ac2.py:21); with
        # check_finite=True train() verifies theta, w and the states after every episode and raises FloatingPointError
        self.check_finite = bool(check_finite)
        self.batch = int(batch)
        # env steps per episode: the reference's loop runs `while num_steps < 15` (mfg_ac2.py:478); BASELINE's synthetic
        # configurations C3 / C5 use T = 40
        self.episode_steps = int(episode_steps)
        if self.episode_steps < 1:
            raise ValueError('episode_steps must be >= 1')
        self._train_bufs = {}      # device buffers of train(), kept between calls (keyed by shape)
        self._force_collective = False   # debug (bench.py --force-dist): take the multi-rank update path with one rank
        self._use_native_rccl = True     # bench.py sets it to False to time the torch.distributed per-episode loop as well
        self._pending = None             # multi-rank rollout mode: (G, lr_critic, lr_actor, reward_acc) of an update not applied yet
        self._w_alt = self._theta_alt = None
        # several GPUs over RCCL: a communicator owned by the HIP library, so that train() issues the one all-reduce per update
        # natively.  Created HERE (a one-time collective hand-shake of the ranks), not inside train().
        self._dist_comm = native_comm(group, self.device) if self.device.type == 'cuda' else None
        self._theta = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._theta_is_array = False
        self.theta = theta
        # critic weights first, start states second: the reference's np.random consumption order (:33, :39)
        self.w = self.init_w(d)
        if pi0 is not None:
            self.mat_pi0 = np.array(pi0, dtype=np.float64)[:, 0:d]
        else:
            if path_to_dir is None:
                path_to_dir = os.getcwd() + '/train_normalized_round2'
            if os.path.isdir(path_to_dir):
                self.init_pi0(path_to_dir=path_to_dir)
            else:
                rs = np.random.RandomState(0)   # synthetic table: Dirichlet(1) rows via '%.3e' text
                m = rs.dirichlet(np.ones(d), size=64)
                self.mat_pi0 = np.array([[float('%.3e' % v) for v in row] for row in m])
        self.num_start_samples = self.mat_pi0.shape[0]
        self._single = False
        self._pi_alpha = None      # state of the last sample_action (hidden coupling of mfg_ac2.py:219-234)
        self._theta_at_sample = self._theta.clone()   # theta the last sample_action ran with (mat_alpha attributes)
        self._rng_step = 0         # Philox step counter (advances once per env step)
        self.trace = None          # set to [] to record theta after every update (parity tests)
        # hand the thread's previous context binding back (the subclasses' constructors launch nothing that reports)
        self._ctx.restore(self.__dict__.pop('_ctx_prev', ops.Context.KEEP))

    # ------------------------------------------------------------------ state on the device
    @property
    def theta(self):
        self._flush_pending()
        v = float(self._theta.cpu()[0])
        return np.array([v]) if self._theta_is_array else v       # ndarray (1,) after the first update (:520)

    @theta.setter
    def theta(self, value):
        self._flush_pending()
        self._theta_is_array = isinstance(value, np.ndarray)
        self._theta.copy_(torch.as_tensor(np.ravel(np.asarray(value, dtype=np.float64))[:1]))

    @property
    def w(self):
        self._flush_pending()
        return self._w.cpu().numpy().reshape(-1, 1).copy()

    @w.setter
    def w(self, value):
        # a multi-rank rollout-mode update may still be pending (its increment belongs to the OLD weights): apply it before
        # the assignment replaces them, like the theta setter -- otherwise the next flush would add it onto the new values
        if getattr(self, '_pending', None) is not None:
            self._flush_pending()
        v = np.ascontiguousarray(np.asarray(value, dtype=np.float64).reshape(-1))
        self._w = torch.as_tensor(v, device=self.device)

    @property
    def mat_pi0(self):
        return self._mat_pi0_host

    @mat_pi0.setter
    def mat_pi0(self, value):
        self._mat_pi0_host = np.array(value, dtype=np.float64)
        self._mat_pi0_dev = torch.as_tensor(np.ascontiguousarray(self._mat_pi0_host, dtype=np.float32),
                                            device=self.device)
        # the start draw is randint(num_start_samples) (mfg_ac2.py:466): keep it in step with the table, whoever sets it
        # (mfg_synthetic.train re-reads the table and updates the count, mfg_synthetic.py:438-440)
        self.num_start_samples = self._mat_pi0_host.shape[0]

    @property
    def mat_alpha(self):
        if self._pi_alpha is None:
            return np.zeros([self.d, self.d])
        a, _ = ops.alpha(self._pi_alpha, self._theta_at_sample, self.shift, want_deriv=False)
        a = a.cpu().numpy()
        return a[0] if self._single else a

    @property
    def mat_alpha_deriv(self):
        if self._pi_alpha is None:
            return np.zeros([self.d, self.d])
        _, ad = ops.alpha(self._pi_alpha, self._theta_at_sample, self.shift, want_alpha=False)
        ad = ad.cpu().numpy()
        return ad[0] if self._single else ad

    # ------------------------------------------------------------------ host helpers (a12)
    def init_w(self, d):
        """U[0,1) column vector of F = d(d+1)/2 + d + 1 weights (mfg_ac2.py:165-176)."""
        num_features = int((d + 1) * d / 2 + d + 1)
        return np.random.rand(num_features, 1)

    def init_pi0(self, path_to_dir, verbose=0):
        """First line of every trend_distribution_day%d.csv, truncated to d columns (mfg_ac2.py:179-208)."""
        rows = []
        num_files = len(os.listdir(path_to_dir))
        for num_day in range(1, 1 + num_files):
            filename = 'trend_distribution_day%d.csv' % num_day
            with open(path_to_dir + '/' + filename, 'r') as f:
                first = f.readline()
            rows.append(list(map(float, first.strip().split(' ')))[0:self.d])
            if verbose:
                print(filename)
        self.mat_pi0 = np.array(rows, dtype=np.float64)

    def reorder(self, list_rows):
        """Order all rows by decreasing popularity of the first row, stable on ties (mfg_ac2.py:58-81)."""
        row1 = list_rows[0]
        order = sorted(range(len(row1)), key=lambda i: row1[i], reverse=True)
        for i in range(len(list_rows)):
            list_rows[i] = [list_rows[i][j] for j in order]
        return list_rows

    def normalize(self, indir='train_round2', outdir='train_normalized_round2', header=True):
        """Row-normalise every file of indir into outdir as '%.3e' space separated text (mfg_ac2.py:116-137)."""
        for filename in os.listdir(os.getcwd() + '/' + indir):
            with open(os.getcwd() + '/' + indir + '/' + filename, 'r') as f:
                if header:
                    f.readline()
                matrix = np.loadtxt(f, delimiter=',')
            matrix = matrix / np.sum(matrix, axis=1, keepdims=True)
            with open(os.getcwd() + '/' + outdir + '/' + filename, 'wb') as f:
                np.savetxt(f, matrix, fmt='%.3e', delimiter=' ')

    # ------------------------------------------------------------------ tensor plumbing
    def _pi_dev(self, pi):
        """(d,) or (B,d) array / tensor -> contiguous fp32 [B,d] device tensor, plus 'single' flag."""
        if isinstance(pi, torch.Tensor):
            t = pi.to(device=self.device, dtype=torch.float32)
        else:
            t = torch.as_tensor(np.asarray(pi, dtype=np.float32), device=self.device)
        single = t.dim() == 1
        if single:
            t = t.unsqueeze(0)
        return t.contiguous(), single

    def _P_dev(self, P):
        if isinstance(P, torch.Tensor):
            t = P.to(device=self.device, dtype=torch.float32)
        else:
            t = torch.as_tensor(np.asarray(P, dtype=np.float32), device=self.device)
        if t.dim() == 2:
            t = t.unsqueeze(0)
        return t.contiguous()

    @staticmethod
    def _out(t, like, single, dtype=np.float64):
        if isinstance(like, torch.Tensor):
            return t[0] if single else t
        a = t.cpu().numpy().astype(dtype)
        return a[0] if single else a

    def _raise_if_not_finite(self, pi, episode):
        ok = torch.isfinite(self._theta).all() & torch.isfinite(self._w).all() & torch.isfinite(pi).all()
        if not bool(ok):
            raise FloatingPointError('non-finite theta / w / state after episode %d (theta = %r)'
                                     % (episode, float(self._theta.cpu()[0])))

    # ------------------------------------------------------------------ checkpoint / resume
    @_with_ctx
    def state_dict(self):
        """Everything a run needs to resume bit for bit: theta, w (fp64, device values copied to the host), the Philox
        step counter and seed, the policy hyper-parameters and the global np.random state (start-state draws).
        The reference keeps no RL checkpoint at all (only CSV logs); `torch.save(obj.state_dict(), path)` is ours."""
        self._flush_pending()
        # tensors and plain Python scalars only, so torch.load(weights_only=True) can read the file: the legacy MT19937
        # state tuple ('MT19937', key[624] uint32, pos, has_gauss, cached_gaussian) is stored field by field
        name, key, pos, has_gauss, cached = np.random.get_state()
        return {'theta': float(self._theta.cpu()[0]), 'theta_is_array': bool(self._theta_is_array),
                'w': self._w.cpu().clone(), 'rng_step': int(self._rng_step), 'seed': int(self.seed),
                'shift': float(self.shift), 'alpha_scale': float(self.alpha_scale), 'd': int(self.d),
                'np_random_key': torch.as_tensor(key.astype(np.int64)), 'np_random_pos': int(pos),
                'np_random_has_gauss': int(has_gauss), 'np_random_cached_gaussian': float(cached)}

    @_with_ctx
    def load_state_dict(self, state, restore_np_random=True):
        if int(state['d']) != int(self.d):
            raise ValueError('checkpoint is for d=%d, object has d=%d' % (state['d'], self.d))
        self._theta.copy_(torch.tensor([state['theta']], dtype=torch.float64))
        self._theta_is_array = bool(state['theta_is_array'])
        self._w.copy_(state['w'].to(self._w.device))
        self._rng_step = int(state['rng_step'])
        self.seed = int(state['seed'])
        self.shift = state['shift']
        self.alpha_scale = state['alpha_scale']
        if restore_np_random:
            if state.get('np_random_key') is not None:
                np.random.set_state(('MT19937', state['np_random_key'].numpy().astype(np.uint32),
                                     int(state['np_random_pos']), int(state['np_random_has_gauss']),
                                     float(state['np_random_cached_gaussian'])))
            elif state.get('np_random_state') is not None:          # round-1 checkpoints: the raw get_state() tuple
                np.random.set_state(tuple(state['np_random_state']))
            else:
                warnings.warn('checkpoint carries no np.random state: the start-state draws of the resumed run will not '
                              'continue the saved stream', RuntimeWarning, stacklevel=2)

    # ------------------------------------------------------------------ a1 + a2
    def _host_gamma(self, pi_dev):
        """Gamma variates for [B,d] states from the global legacy np.random stream, in the reference's order:
        trajectory by trajectory, row by row, one vector draw of length d per row (mfg_ac2.py:238-242)."""
        a, _ = ops.alpha(pi_dev, self._theta, self.shift, want_deriv=False)
        a = a.cpu().numpy()
        B, d = a.shape[0], self.d
        y = np.empty((B, d, d))
        for b in range(B):
            for i in range(d):
                y[b, i] = np.random.gamma(shape=a[b, i, :] * self.alpha_scale, scale=1)
        return torch.as_tensor(y.astype(np.float32), device=self.device)

    def _sample(self, pi_dev, traj_offset=0, snapshot=True):
        self._pi_alpha = pi_dev
        if snapshot:                       # theta at sampling time, for the mat_alpha / mat_alpha_deriv attributes
            self._theta_at_sample = self._theta.clone()
        if self.rng == 'numpy':
            return ops.dirichlet_from_gamma(self._host_gamma(pi_dev))
        P = ops.sample_dirichlet(pi_dev, self._theta, self.shift, self.alpha_scale, self.seed, self._rng_step,
                                 traj_offset, precision=self.precision)
        self._rng_step += 1
        return P

    @_with_ctx
    def sample_action(self, pi):
        """P ~ prod_i Dirichlet(alpha_i. * alpha_scale); (d,) -> (d,d), (B,d) -> (B,d,d)  (mfg_ac2.py:211-254)."""
        pi_dev, single = self._pi_dev(pi)
        self._single = single
        return self._out(self._sample(pi_dev), pi, single)

    # ------------------------------------------------------------------ a3 - a5, a7
    @_with_ctx
    def calc_reward(self, P, pi, d=None):
        """R = sum_i pi_i sum_j P_ij^2 (pi_j - pi_i); shape (1,) for one trajectory (mfg_ac2.py:257-287)."""
        pi_dev, single = self._pi_dev(pi)
        _, r = ops.step_given_P(pi_dev, self._P_dev(P), reward_kind=self.reward_kind)
        if isinstance(pi, torch.Tensor):
            return r
        return r.cpu().numpy().astype(np.float64)          # (1,) when single, like the reference

    @_with_ctx
    def transition(self, P, pi):
        """pi' = P^T pi (mfg_ac2.py:497)."""
        pi_dev, single = self._pi_dev(pi)
        pn, _ = ops.step_given_P(pi_dev, self._P_dev(P), want_reward=False)
        return self._out(pn, pi, single)

    @_with_ctx
    def calc_features(self, pi):
        pi_dev, single = self._pi_dev(pi)
        return self._out(ops.features(pi_dev), pi, single)

    @_with_ctx
    def calc_value(self, pi):
        pi_dev, single = self._pi_dev(pi)
        v = ops.value(pi_dev, self._w)
        if isinstance(pi, torch.Tensor):
            return v
        return v.cpu().numpy()                               # (1,) when single (mfg_ac2.py:311)

    @_with_ctx
    def calc_gradient_vectorized(self, P, pi):
        """Score of the product-Dirichlet policy w.r.t. theta (mfg_ac2.py:347-381).  Like the reference it
        uses the concentrations of the LAST sample_action call when there is one (hidden state), and
        replaces zeros of a NumPy ``P`` by 1e-100 in place (:369)."""
        pi_dev, single = self._pi_dev(pi)
        pa = self._pi_alpha if (self._pi_alpha is not None and self._pi_alpha.shape == pi_dev.shape) else pi_dev
        g = ops.score(pa, self._P_dev(P), self._theta, self.shift, precision=self.precision)
        if isinstance(P, np.ndarray) and P.dtype == np.float64:
            P[P == 0] = 1e-100
        if isinstance(pi, torch.Tensor):
            return g[0] if single else g
        g = g.cpu().numpy()
        return float(g[0]) if single else g

    def _gradient_inputs(self, P, pi):
        """(alpha, alpha', ln P) as fp64 device tensors [B,d,d] for the two loop-form score variants below: mfg_alpha for
        the concentrations of the LAST sample_action state (the reference's hidden mat_alpha / mat_alpha_deriv), the
        stored probabilities widened from the fp32 the device holds; zeros of P -> 1e-100 like calc_gradient_vectorized
        (the loop forms of the reference do not floor them and would return -inf)."""
        pi_dev, single = self._pi_dev(pi)
        pa = self._pi_alpha if (self._pi_alpha is not None and self._pi_alpha.shape == pi_dev.shape) else pi_dev
        a, ad = ops.alpha(pa, self._theta, self.shift)
        Pd = self._P_dev(P).double()
        lnP = torch.log(torch.where(Pd == 0, torch.full_like(Pd, 1e-100), Pd))
        return a, ad, lnP, single

    def _gradient_out(self, g, pi, single):
        if isinstance(pi, torch.Tensor):
            return g[0] if single else g
        g = g.cpu().numpy()
        return float(g[0]) if single else g

    @_with_ctx
    def calc_gradient_basic(self, P, pi):
        """The reference's first loop form (mfg_ac2.py:384-400): per row i THREE separate sums, added in this order --
        -sum_j psi(alpha_ij) alpha'_ij, then psi(sum_j alpha_ij) sum_j alpha'_ij, then sum_j ln(P_ij) alpha'_ij.
        An evaluation path of its OWN (mfg_alpha + torch.special.digamma + log on the device, fp64; no score kernel), so
        that the reference's three-way self-check (test2.py:105-121) compares independent computations here too."""
        a, ad, lnP, single = self._gradient_inputs(P, pi)
        t1 = -(torch.special.digamma(a) * ad).sum(-1)
        t2 = torch.special.digamma(a.sum(-1)) * ad.sum(-1)
        t3 = (lnP * ad).sum(-1)
        return self._gradient_out(((t1 + t2) + t3).sum(-1), pi, single)

    @_with_ctx
    def calc_gradient(self, P, pi):
        """The reference's second loop form (mfg_ac2.py:402-438): per element (-psi(alpha_ij) + psi(sum_j alpha_ij) +
        ln P_ij) alpha'_ij, summed over the matrix.  Its own device evaluation like calc_gradient_basic."""
        a, ad, lnP, single = self._gradient_inputs(P, pi)
        mult = torch.special.digamma(a.sum(-1, keepdim=True))
        return self._gradient_out((((-torch.special.digamma(a) + mult) + lnP) * ad).sum((-2, -1)), pi, single)

    # ------------------------------------------------------------------ logging (mfg_ac2.py:441-445)
    def train_log(self, vector, filename, str_format):
        with open(filename, 'a') as f:
            np.asarray(vector).tofile(f, sep=',', format=str_format)
            f.write('\n')

    # ------------------------------------------------------------------ a9: train
    def _device_draw(self):
        """Whether train() draws the start states on the device.  The reference draws `np.random.randint(num_start)` per
        episode (mfg_ac2.py:466); that is kept for the runs that retrace it -- batch 1, or rng='numpy'.  Batched Philox
        runs (no reference counterpart) draw idx_b from the action sampler's counter-based generator, keyed by (seed,
        episode's first Philox step, GLOBAL trajectory id): no host RNG, no index upload, no broadcast between ranks."""
        return self.rng == 'philox' and self.batch > 1

    def _draw_start(self, shard):
        """Host draw of the start-state indices (batch 1 / rng='numpy'): the GLOBAL index vector comes from the
        process-global legacy np.random stream (one scalar draw at batch 1, the reference's :466); with several ranks,
        rank 0's draw is broadcast, so the ranks need not share a host seed, and every rank keeps its shard of it."""
        if self.batch == 1:
            idx = np.array([np.random.randint(self.num_start_samples)])      # the reference's single draw (:466)
        else:
            idx = np.random.randint(self.num_start_samples, size=self.batch)
        if shard.world > 1:
            idx = broadcast_start_indices(idx, self.group, self.device)
            idx = idx[shard.traj_offset:shard.traj_offset + shard.local_batch]
        return torch.as_tensor(idx.astype(np.int32), device=self.device)

    @_with_ctx
    def train(self, num_episodes=4000, gamma=1, constant=0, lr_critic=0.1, lr_actor=0.001, consecutive=100,
              file_theta='results/theta.csv', file_pi='results/pi.csv', file_reward='results/reward.csv',
              write_file=0, write_all=0, *, first_episode=0):
        """Actor-critic training (mfg_ac2.py:448-539) over ``batch`` lock-step trajectories.
        batch=1, rng='numpy', update_every='step' retraces the reference's seeded run.
        first_episode: episode number the 1/(episode+1) learning-rate schedule starts from (resume after
        load_state_dict)."""
        d, T = self.d, self.episode_steps
        shard = current_shard(self.batch, self.group)
        Bl = shard.local_batch
        if shard.world > self.batch:
            raise ValueError('batch=%d is smaller than the world size %d: every rank needs a trajectory'
                             % (self.batch, shard.world))
        F = ops.num_features(d)
        # device buffers: allocated (and the workspace zeroed) once per shape, reused by later train() calls -- the outer
        # loops of the IRL class and resumed runs call train() many times
        key = ('train', Bl, T, d, str(self.device))
        bufs = self._train_bufs.get(key)
        if bufs is None:
            self._train_bufs.clear()
            bufs = {'G': torch.zeros(F + 3, dtype=torch.float64, device=self.device),
                    'ws': ops.workspace(Bl * T, d, self.device)}
            self._train_bufs[key] = bufs
        G, ws = bufs['G'], bufs['ws']
        # mean reward per update, one entry per episode (step mode: summed over the episode's T updates = the mean episode
        # return; rollout mode: the mean over the B*T transitions of the one update, times T below)
        # (kept per instance and re-zeroed: a fresh allocation per call is host time in front of the first launch; an update an
        #  unwound call left pending books its reward into this buffer: apply it before the entries are reset)
        self._flush_pending()
        ep_reward = bufs.get('ep_reward')
        if ep_reward is None or ep_reward.numel() < max(num_episodes, 1):
            ep_reward = bufs['ep_reward'] = torch.zeros(max(num_episodes, 1), dtype=torch.float64, device=self.device)
        else:
            ep_reward.zero_()
        ep_base = ep_reward.data_ptr()
        # a pending multi-rank update (self._pending) carries a raw address into ep_reward: the tensor must outlive this call
        # if train() unwinds with the update still pending (flushed later by a read of theta / w / state_dict)
        self._ep_reward_keepalive = ep_reward
        ret_scale = float(T) if self.update_every == 'rollout' else 1.0
        window_start = 0
        pi = None
        device_draw = self._device_draw()
        # per-step updates on one GPU: the whole episode is issued by native code (mfg_train_episode)
        native_episode = (self.update_every == 'step' and self.rng == 'philox' and shard.world == 1
                          and not self._force_collective and self.trace is None and not write_all)
        ebufs = None
        if native_episode:
            if 'episode' not in bufs:
                bufs['episode'] = ops.episode_buffers(Bl, d, self.device)
            ebufs = bufs['episode']
        fused_rollout = self.update_every == 'rollout' and self.rng == 'philox'
        if fused_rollout:
            if 'rollout' not in bufs:
                bufs['rollout'] = {'pi_traj': torch.empty(Bl, T + 1, d, dtype=torch.float32, device=self.device),
                                   'pi_last': torch.empty(Bl, d, dtype=torch.float32, device=self.device),
                                   'reward': torch.empty(Bl, T, dtype=torch.float32, device=self.device),
                                   'delta': torch.empty(Bl, T, dtype=torch.float64, device=self.device),
                                   'g': torch.empty(Bl, T, dtype=torch.float64, device=self.device)}
            rbufs = bufs['rollout']
        # one GPU, start states drawn on the device, nothing for the host to do between two episodes: ALL episodes up to
        # the next report are issued by one native call (mfg_train_rollouts / mfg_train_episodes: the draw, the learning-rate
        # schedule and the episode loop itself run without the interpreter)
        multi = shard.world > 1 or self._force_collective
        native_loop = (device_draw and not multi and (fused_rollout or native_episode) and self.trace is None
                       and not write_all and not self.check_finite)
        # several GPUs, one update per episode: the same native loop with the all-reduce issued by the library (RCCL)
        dist_comm = None
        if multi and fused_rollout and device_draw and self.trace is None and not write_all and not self.check_finite:
            dist_comm = self._dist_comm if self._use_native_rccl else None
            if dist_comm is None and self._force_collective and self._use_native_rccl:      # debug (one rank): made on first use
                dist_comm = native_comm(self.group, self.device, allow_single=True)
            if dist_comm is not None and (self._w_alt is None or self._w_alt.shape != self._w.shape):
                self._w_alt, self._theta_alt = torch.empty_like(self._w), torch.empty_like(self._theta)
        if native_episode and device_draw:
            if 'pi_ep' not in bufs:
                bufs['pi_ep'] = torch.empty(Bl, d, dtype=torch.float32, device=self.device)
            pi_ep = bufs['pi_ep']
        # reports (every `consecutive` episodes): in the native loop their values are read back behind the NEXT chunk of
        # episodes; a subclass hook that reads the live parameters (mfg_synthetic logs w) keeps the immediate form
        defer_reports = (native_loop or dist_comm is not None) and type(self)._train_log_extra is actor_critic._train_log_extra
        # nobody consumes the reports (nothing printed, nothing logged): no snapshot, no copy, no event, no pinned buffer --
        # and the native loops need not stop at the reporting episodes: ONE native call issues the whole train() call
        silent = not self.verbose and not write_file
        report = None
        episode = 0
        while episode < num_episodes:
            if native_loop or dist_comm is not None:
                # episodes episode .. last, `last` = the next reporting episode (episode % consecutive == 0) or the final one
                last = episode if episode % consecutive == 0 else (episode // consecutive + 1) * consecutive
                last = num_episodes - 1 if silent else min(last, num_episodes - 1)
                k = last - episode + 1
                if dist_comm is not None:
                    self._flush_pending()
                    try:
                        ops.train_rollouts_dist(dist_comm, self._mat_pi0_dev, T, k, episode + first_episode, constant == 1,
                                                self._theta, self._w, self._theta_alt, self._w_alt, self.shift, self.alpha_scale,
                                                gamma, G, ws, rbufs, lr_critic, lr_actor, reward_kind=self.reward_kind,
                                                seed=self.seed, first_step=self._rng_step, traj_offset=shard.traj_offset,
                                                reward_acc=ep_base + 8 * episode, precision=self.precision)
                    except L.MfgError as exc:
                        if exc.code == L.ECOMM:
                            # the library aborted its communicator inside the call (a launch or collective failed): the handle
                            # is dead -- forget it everywhere, so that a later train() falls back to torch.distributed instead
                            # of calling into freed memory
                            forget_native_comm(dist_comm)
                            self._dist_comm = None
                        raise
                    pi = rbufs['pi_last']
                elif fused_rollout:
                    ops.train_rollouts(self._mat_pi0_dev, T, k, episode + first_episode, constant == 1, self._theta, self.shift,
                                       self.alpha_scale, self._w, gamma, G, ws, rbufs, lr_critic, lr_actor,
                                       reward_kind=self.reward_kind, seed=self.seed, first_step=self._rng_step,
                                       traj_offset=shard.traj_offset, reward_acc=ep_base + 8 * episode,
                                       precision=self.precision)
                    pi = rbufs['pi_last']
                else:
                    ops.train_episodes(self._mat_pi0_dev, pi_ep, T, k, episode + first_episode, constant == 1, self._theta,
                                       self.shift, self.alpha_scale, self._w, gamma, lr_critic, lr_actor, G, ws, ebufs,
                                       reward_kind=self.reward_kind, seed=self.seed, first_step=self._rng_step,
                                       traj_offset=shard.traj_offset, reward_acc=ep_base + 8 * episode,
                                       precision=self.precision)
                    pi = pi_ep
                self._rng_step += k * T
                self._theta_is_array = True
                episode = last
            else:
                pi = self._train_one_episode(episode, shard, device_draw, native_episode, fused_rollout, G, ws, ebufs,
                                             rbufs if fused_rollout else None, ep_base + 8 * episode, gamma, constant,
                                             lr_critic, lr_actor, first_episode, write_all)
            if self.check_finite or episode % consecutive == 0 or episode == num_episodes - 1:
                self._flush_pending()
            if self.check_finite:
                self._raise_if_not_finite(pi, episode)
            if report is not None:
                # the previous report: its values were copied out asynchronously BEFORE the chunk above was enqueued, so
                # waiting for them does not drain the launch stream (the GPU keeps working on that chunk meanwhile)
                self._emit_report(report, consecutive, write_file, file_theta, file_pi, file_reward)
                report = None
            if episode % consecutive == 0 and not silent:
                # the reference divides the sum over the window by `consecutive` even at episode 0 (:530-534)
                report = self._snapshot_report(ep_reward, window_start, episode, pi, ret_scale)
                window_start = episode + 1
                if not defer_reports:
                    self._emit_report(report, consecutive, write_file, file_theta, file_pi, file_reward)
                    report = None
            episode += 1
        if report is not None:
            self._emit_report(report, consecutive, write_file, file_theta, file_pi, file_reward)
        self._last_pi = pi
        self._check_status()

    def _snapshot_report(self, ep_reward, window_start, episode, pi, ret_scale):
        """Values of a report (theta, the first trajectory's state, the window's reward sum) copied to pinned host memory
        without waiting: device-side snapshots, non-blocking copies, one event."""
        # two pinned buffers per instance, used in turn (a deferred report is still unread when the next one is taken);
        # allocating pinned memory per report costs a hipHostMalloc each time
        pool = self.__dict__.setdefault('_report_host', [None, None, 0])
        slot = pool[2] & 1
        pool[2] += 1
        if pool[slot] is None or pool[slot].numel() != self.d + 2:
            pool[slot] = torch.empty(self.d + 2, dtype=torch.float64, pin_memory=True)
        host = pool[slot]
        dev = torch.cat([self._theta, ep_reward[window_start:episode + 1].sum().reshape(1) * ret_scale, pi[0].double()])
        host.copy_(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return {'host': host, 'event': ev, 'dev': dev, 'theta_is_array': self._theta_is_array}

    def _emit_report(self, report, consecutive, write_file, file_theta, file_pi, file_reward):
        report['event'].synchronize()
        h = report['host'].numpy()
        theta = np.array([h[0]]) if report['theta_is_array'] else float(h[0])
        reward_avg = float(h[1]) / consecutive
        pi_host = h[2:].astype(np.float32).astype(np.float64)       # (the state is fp32 on the device)
        if self.verbose:
            print('Theta\n', theta)
            print('pi\n', pi_host)
            print('Average reward during previous %d episodes: ' % consecutive, str(reward_avg))
        if write_file:
            self.train_log(np.ravel(theta), file_theta, '%.5e')
            self.train_log(pi_host, file_pi, '%.3e')
            self.train_log(np.array([reward_avg]), file_reward, '%.3e')
            self._train_log_extra()

    def _flush_pending(self):
        """Apply the update a multi-rank rollout-mode episode left pending (see _train_one_episode): before anything reads
        theta / w on the host, and after the last episode."""
        if self._pending is not None:
            G, lc, la, racc = self._pending
            prev = self._ctx.bind_scoped()          # (reached from the unbound property reads of theta / w too)
            try:
                ops.apply_update(G, self.d, lc, la, self._w, self._theta, racc)
            finally:
                self._ctx.restore(prev)
            self._pending = None

    def _train_one_episode(self, episode, shard, device_draw, native_episode, fused_rollout, G, ws, ebufs, rbufs, reward_acc,
                           gamma, constant, lr_critic, lr_actor, first_episode, write_all):
        """One episode of train() issued from Python: the multi-rank path (an all-reduce sits between the batch sums and
        the update), runs that record a trace / check finiteness / write every step, and the reference-RNG modes.
        reward_acc: device address of this episode's entry of the return accumulator.  Returns the final states."""
        d, T = self.d, self.episode_steps
        Bl = shard.local_batch
        if write_all:
            with open('temp.csv', 'a') as f:
                f.write('Episode %d \n\n' % episode)
        sc, sa = lr_scales(episode + first_episode, constant == 1)
        if fused_rollout:
            # start states (drawn in the kernel, or gathered from the host draw) + fused T-step rollout + batch sums
            # (+ the update itself on one GPU): 2-3 launches
            idx = None if device_draw else self._draw_start(shard)
            single = shard.world == 1 and not self._force_collective
            if not single and self.trace is None:
                # several ranks: rollout | sums | all-reduce -- and nothing else.  The update of THIS episode is left pending
                # and applied by the next episode's rollout kernel while it stages its weights (mfg_train_rollout_deferred;
                # parameters ping-pong between two buffers); _flush_pending applies the last one.
                if self._w_alt is None or self._w_alt.shape != self._w.shape:
                    self._w_alt, self._theta_alt = torch.empty_like(self._w), torch.empty_like(self._theta)
                ops.train_rollout_deferred(self._mat_pi0_dev, idx, T, self._theta, self._w, self._pending, self._theta_alt,
                                           self._w_alt, self.shift, self.alpha_scale, gamma, G, ws, rbufs,
                                           reward_kind=self.reward_kind, seed=self.seed, first_step=self._rng_step,
                                           traj_offset=shard.traj_offset, precision=self.precision)
                if self._pending is not None:
                    self._theta, self._theta_alt = self._theta_alt, self._theta
                    self._w, self._w_alt = self._w_alt, self._w
                self._rng_step += T
                all_reduce_gradients_(G, self.group, self._force_collective)       # the ONE exchange of an update
                self._pending = (G, lr_critic * sc, lr_actor * sa, reward_acc)
                self._theta_is_array = True
                return rbufs['pi_last']
            ops.train_rollout(self._mat_pi0_dev, idx, T, self._theta, self.shift, self.alpha_scale, self._w, gamma, G,
                              ws, rbufs, lr_critic * sc, lr_actor * sa, apply=single, reward_kind=self.reward_kind,
                              seed=self.seed, first_step=self._rng_step, traj_offset=shard.traj_offset,
                              reward_acc=reward_acc, precision=self.precision)
            self._rng_step += T
            if not single:
                all_reduce_gradients_(G, self.group, self._force_collective)       # the ONE exchange of an update
                ops.apply_update(G, d, lr_critic * sc, lr_actor * sa, self._w, self._theta, reward_acc)
            self._theta_is_array = True
            if self.trace is not None:
                self.trace.append(float(self._theta.cpu()[0]))
            return rbufs['pi_last']
        if device_draw:
            _, pi = ops.draw_start(self._mat_pi0_dev, Bl, self.seed, self._rng_step, shard.traj_offset)
        else:
            pi = ops.gather_start(self._mat_pi0_dev, self._draw_start(shard))
        if native_episode:
            ops.train_episode(pi, T, self._theta, self.shift, self.alpha_scale, self._w, gamma, lr_critic * sc,
                              lr_actor * sa, G, ws, ebufs, reward_kind=self.reward_kind, seed=self.seed,
                              first_step=self._rng_step, traj_offset=shard.traj_offset,
                              reward_acc=reward_acc, precision=self.precision)
            self._rng_step += T
            self._theta_is_array = True
            return pi
        for step in range(T):
            if self.rng == 'philox':
                out = ops.rollout(pi, 1, self._theta, self.shift, self.alpha_scale, w=self._w, gamma=gamma,
                                  reward_kind=self.reward_kind, seed=self.seed, first_step=self._rng_step,
                                  traj_offset=shard.traj_offset, td=True, G=G, ws=ws, precision=self.precision,
                                  accumulate=(self.update_every == 'rollout' and step > 0))
                self._rng_step += 1
                pi_next = out['pi_last']
            else:
                P = self._sample(pi, shard.traj_offset, snapshot=False)
                if write_all:
                    self._write_all(pi, P, step + 1)
                pi_next, r = ops.step_given_P(pi, P, reward_kind=self.reward_kind)
                ops.td_pg_accumulate(pi, pi_next, P, r, self._w, self._theta, self.shift, gamma, G=G, ws=ws,
                                     precision=self.precision,
                                     accumulate=(self.update_every == 'rollout' and step > 0))
            if self.update_every == 'step':
                all_reduce_gradients_(G, self.group, self._force_collective)
                ops.apply_update(G, d, lr_critic * sc, lr_actor * sa, self._w, self._theta, reward_acc)
                self._theta_is_array = True
                if self.trace is not None:
                    self.trace.append(float(self._theta.cpu()[0]))
            pi = pi_next
        if self.update_every == 'rollout':
            all_reduce_gradients_(G, self.group, self._force_collective)
            ops.apply_update(G, d, lr_critic * sc, lr_actor * sa, self._w, self._theta, reward_acc)
            self._theta_is_array = True
            if self.trace is not None:
                self.trace.append(float(self._theta.cpu()[0]))
        return pi

    def status(self, synchronize=True) -> int:
        """Bits of THIS instance's status word (0 = healthy; include/mfg_hip.h mfg_ctx_status)."""
        return self._ctx.status(synchronize)

    def clear_status(self):
        """Reset this instance's sticky status word after a reported numeric-range condition (mfg_ctx_clear_status)."""
        self._ctx.clear_status()

    def _check_status(self):
        """Raise MfgError if a launch of this run reported a numeric-range condition (mixed-precision sampling with
        theta outside its range, include/mfg_hip.h mfg_status) -- the end-of-train check; during a run the next sampling
        launch after the condition is refused by the library itself.  The word belongs to this instance's context: another
        instance's diverged run neither raises here nor stops this one's launches."""
        if self.precision == 'mixed' and self.rng == 'philox' and self._ctx.status(synchronize=True):
            L.check(L.lib().mfg_ctx_status(self._ctx._ptr, None), 'train')

    def _train_log_extra(self):
        """Hook for the variants' additional per-report log lines (mfg_synthetic.py:522 logs w)."""

    def _write_all(self, pi, P, num_steps):
        with open('temp.csv', 'ab') as f:
            np.savetxt(f, np.array(['num_steps = %d' % num_steps]), fmt='%s')
            np.savetxt(f, np.array(['distribution']), fmt='%s')
            np.savetxt(f, pi[0].cpu().numpy().reshape(1, self.d), delimiter=',', fmt='%.6f')
            np.savetxt(f, np.array(['Action']), fmt='%s')
            np.savetxt(f, P[0].cpu().numpy(), delimiter=',', fmt='%.3f')

    # ------------------------------------------------------------------ a10, a11: evaluation
    def JSD(self, P, Q):
        """Jensen-Shannon divergence, zeros -> 1e-100 (mfg_ac2.py:546-563); scalar for 1-D inputs."""
        p, single = self._pi_dev(P)
        q, _ = self._pi_dev(Q)
        if isinstance(P, np.ndarray) and P.dtype == np.float64:
            P[P == 0] = 1e-100                               # the reference mutates its arguments (:556-557)
        if isinstance(Q, np.ndarray) and Q.dtype == np.float64:
            Q[Q == 0] = 1e-100
        out = ops.jsd(p, q)
        if isinstance(P, torch.Tensor):
            return out[0] if single else out
        out = out.cpu().numpy()
        return float(out[0]) if single else out

    @_with_ctx
    def generate_trajectory(self, pi0, total_hours):
        """Rows pi^0 .. pi^{total_hours-1} under the current policy (mfg_ac2.py:566-592);
        (d,) -> (total_hours, d), (B,d) -> (B, total_hours, d)."""
        pi_dev, single = self._pi_dev(pi0)
        T = total_hours - 1
        if self.rng == 'philox' and T >= 1:
            out = ops.rollout(pi_dev, T, self._theta, self.shift, self.alpha_scale, seed=self.seed,
                              first_step=self._rng_step, td=False, reward_kind=self.reward_kind, precision=self.precision)
            self._rng_step += T
            traj = out['pi_traj']
        else:
            # rng='numpy': the reference generates each trajectory to completion before the next one starts
            # (evaluate, mfg_ac2.py:629-633 -> :585-590), so the legacy np.random stream is consumed trajectory-major
            trajs = []
            for b in range(pi_dev.shape[0]):
                pi = pi_dev[b:b + 1].contiguous()
                rows = [pi]
                for _ in range(T):
                    P = self._sample(pi)
                    pi, _r = ops.step_given_P(pi, P, want_reward=False)
                    rows.append(pi)
                trajs.append(torch.cat(rows, dim=0))
            traj = torch.stack(trajs, dim=0)
        return self._out(traj, pi0, single)

    _EVAL_HEADER = ('theta,shift,alpha_scale,mean_l1_final,std_l1_final,mean_l1_mean,std_l1_mean,'
                    'mean_JSD_final,std_JSD_final,mean_JSD_mean,std_JSD_mean\n')

    def _load_empirical(self, indir, d, episode_length):
        """Every file of cwd/indir as [N, L, d] (mfg_ac2.py:612-626): fp64 for the L1 metric, fp32 for the kernels."""
        path_to_dir = os.getcwd() + '/' + indir
        emp = []
        for filename in os.listdir(path_to_dir):
            with open(path_to_dir + '/' + filename, 'r') as f:
                emp.append(np.loadtxt(f, delimiter=' ')[:, 0:d])
        emp = np.array(emp)[:, :episode_length]
        return (torch.as_tensor(emp, dtype=torch.float64, device=self.device),
                torch.as_tensor(emp.astype(np.float32), device=self.device))

    def _eval_metrics_dev(self, emp64, emp32):
        """The eight metrics of evaluate() for the CURRENT policy, computed on the device (fp64 tensor [8], no host
        round trip): all test trajectories are generated in one launch, L1 per step by torch, JSD by mfg_jsd."""
        N, Lh, d = emp32.shape
        gen = self.generate_trajectory(emp32[:, 0].contiguous(), Lh)                       # [N, L, d] device tensor
        diff = (emp64 - gen.double()).abs().sum(-1)                                       # L1 per step
        jsd = ops.jsd(emp32.reshape(N * Lh, d).contiguous(), gen.reshape(N * Lh, d).contiguous()).view(N, Lh)
        std = lambda t: t.std(unbiased=False)                                             # numpy's default ddof = 0
        return torch.stack([diff[:, -1].mean(), std(diff[:, -1]), diff.mean(1).mean(), std(diff.mean(1)),
                            jsd[:, -1].mean(), std(jsd[:, -1]), jsd.mean(1).mean(), std(jsd.mean(1))])

    @_with_ctx
    def evaluate(self, theta=8.86349, shift=0.5, alpha_scale=1e4, d=21, episode_length=16,
                 indir='test_normalized_round2', outfile='eval_mfg_round2/test_eval_fixed_reward.csv',
                 write_header=0):
        """L1 / JSD of generated vs empirical trajectories over every file of indir (mfg_ac2.py:595-670).
        All test trajectories are generated in one batched launch; the metrics are reduced on the device."""
        self.theta = theta
        self.shift = shift
        self.alpha_scale = alpha_scale
        self.d = d
        emp64, emp32 = self._load_empirical(indir, d, episode_length)
        res = [float(v) for v in self._eval_metrics_dev(emp64, emp32).cpu()]
        with open(outfile, 'a') as f:
            if write_header:
                f.write(self._EVAL_HEADER)
            f.write('%f,%f,%f,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e\n' % ((theta, shift, alpha_scale) + tuple(res)))
        return res[0], res[2], res[4], res[6]

    @_with_ctx
    def gridsearch(self, theta_range, shift_range, alpha_range, indir, outfile):
        """Sweep (theta, shift, alpha_scale) and keep the best of each metric (mfg_ac2.py:673-689).  The test files
        are read once, every grid point is evaluated on the device back to back (one rollout + one JSD launch per
        point, no host synchronisation in the sweep); the CSV lines evaluate() would have appended and the argmin
        scan of the reference are produced from one final copy.  Same numbers as calling evaluate() per point."""
        emp64, emp32 = self._load_empirical(indir, self.d, 16)
        points, rows = [], []
        for theta in theta_range:
            for shift in shift_range:
                for alpha_scale in alpha_range:
                    if self.verbose:
                        print('Theta %f, shift %f, alpha %d' % (theta, shift, alpha_scale))
                    self.theta = theta
                    self.shift = shift
                    self.alpha_scale = alpha_scale
                    points.append((theta, shift, alpha_scale))
                    rows.append(self._eval_metrics_dev(emp64, emp32))
        list_tuples = [[100, 0, 0, 0], [100, 0, 0, 0], [100, 0, 0, 0], [100, 0, 0, 0]]
        if not rows:
            return list_tuples
        table = torch.stack(rows).cpu().numpy()                                # the only synchronisation of the sweep
        with open(outfile, 'a') as f:
            for (theta, shift, alpha_scale), res in zip(points, table):
                f.write('%f,%f,%f,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e\n' % ((theta, shift, alpha_scale) + tuple(res)))
                result = (res[0], res[2], res[4], res[6])
                for idx in range(4):
                    if result[idx] <= list_tuples[idx][0]:
                        list_tuples[idx] = [float(result[idx]), theta, shift, alpha_scale]
        if self.verbose:
            print(list_tuples)
        return list_tuples
