"""Drop-in for the reference's ``mfg_synthetic.actor_critic`` (mfg_synthetic.py:24): the same actor-critic
as mfg_ac2 with
  * reward R = -1/2 sum_i pi_i ||P_i||^2                          (mfg_synthetic.py:249-265),
  * ``generate_trajectory`` returning (trajectory, actions)       (:549-578),
  * ``JSD`` treating every entry <= 0 as 1e-100                    (:528-547),
  * the backward-equation consistency checks ``evaluate_synthetic`` / ``evaluate_synthetic_JSD``
    (V^n = r + P V^{n+1}, :741-899), here one batched reverse-time scan over all start states.
Constructor defaults are the reference's (theta=10, shift=0, alpha_scale=100, d=21); start states come from
``pi0=`` or from cwd/train_normalized/trend_distribution_day%d_reordered.csv (:181, :439) when present.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from .mfg_ac2 import _with_ctx, actor_critic as _base


class actor_critic(_base):

    def __init__(self, theta=10, shift=0, alpha_scale=100, d=21, **kw):
        kw.setdefault('reward', 'synthetic')
        if kw.get('pi0') is None and kw.get('path_to_dir') is None and os.path.isdir(os.getcwd() + '/train_normalized'):
            kw['path_to_dir'] = os.getcwd() + '/train_normalized'
        super().__init__(theta=theta, shift=shift, alpha_scale=alpha_scale, d=d, **kw)

    def init_pi0(self, path_to_dir, verbose=0):
        """First line of trend_distribution_day%d_reordered.csv, truncated to d (mfg_synthetic.py:169-199)."""
        rows = []
        for num_day in range(1, 1 + len(os.listdir(path_to_dir))):
            with open(path_to_dir + '/trend_distribution_day%d_reordered.csv' % num_day, 'r') as f:
                rows.append(list(map(float, f.readline().strip().split(' ')))[0:self.d])
        self.mat_pi0 = np.array(rows, dtype=np.float64)

    @_with_ctx
    def train(self, num_episodes=4000, gamma=1, constant=0, lr_critic=0.1, lr_actor=0.001, consecutive=100,
              file_theta='results_syn/theta.csv', file_pi='results_syn/pi.csv', file_reward='results_syn/reward.csv',
              file_w='results_syn/w.csv', write_file=0, write_all=0, **kw):
        """mfg_synthetic.py:426-522: the mfg_ac2 loop with the synthetic reward, the `results_syn/` log defaults, the start
        states re-read from cwd/train_normalized when that directory exists (:438-440), and one more log line per
        report: w (`file_w`, '%.5e', :522)."""
        if os.path.isdir(os.getcwd() + '/train_normalized'):
            self.init_pi0(path_to_dir=os.getcwd() + '/train_normalized')
        self._file_w = file_w
        try:
            return super().train(num_episodes=num_episodes, gamma=gamma, constant=constant, lr_critic=lr_critic,
                                 lr_actor=lr_actor, consecutive=consecutive, file_theta=file_theta, file_pi=file_pi,
                                 file_reward=file_reward, write_file=write_file, write_all=write_all, **kw)
        finally:
            self._file_w = None

    def _train_log_extra(self):
        if getattr(self, '_file_w', None):
            self.train_log(np.ravel(self.w), self._file_w, '%.5e')

    @_with_ctx
    def calc_reward_vector(self, P):
        """v_i = -1/2 ||P_i||^2 (mfg_synthetic.py:726-738) via the backward kernel on a 1-step sequence."""
        Pd = self._P_dev(P)                               # [B,d,d]
        V, _, _ = ops.backward_value(Pd.unsqueeze(1).contiguous(), want_jsd=False)
        v = V[:, 0].cpu().numpy()
        return v[0] if np.asarray(P).ndim == 2 else v

    @_with_ctx
    def JSD(self, P, Q):
        """Entries <= 0 count as 1e-100 (mfg_synthetic.py:540-541; the base class only replaces exact zeros)."""
        P = np.array(P, dtype=np.float64); Q = np.array(Q, dtype=np.float64)
        P[P <= 0] = 0.0; Q[Q <= 0] = 0.0
        return super().JSD(P, Q)

    @_with_ctx
    def generate_trajectory(self, pi0, total_hours):
        """(mat_trajectory [total_hours,d], array_actions [total_hours-1,d,d]) (mfg_synthetic.py:549-578);
        batched for (B,d) input."""
        pi_dev, single = self._pi_dev(pi0)
        T = total_hours - 1
        if self.rng == 'philox':
            out = ops.rollout(pi_dev, T, self._theta, self.shift, self.alpha_scale, seed=self.seed,
                              first_step=self._rng_step, td=False, write_P=True, reward_kind=self.reward_kind,
                              precision=self.precision)
            self._rng_step += T
            traj, acts = out['pi_traj'], out['P']
        else:
            rows, As, pi = [pi_dev], [], pi_dev
            for _ in range(T):
                P = self._sample(pi)
                pi, _r = ops.step_given_P(pi, P, want_reward=False)
                rows.append(pi); As.append(P)
            traj, acts = torch.stack(rows, dim=1), torch.stack(As, dim=1)
        return self._out(traj, pi0, single), self._out(acts, pi0, single)

    def _evaluate_synthetic(self, day_first, day_last, jsd):
        pi0 = self.mat_pi0[day_first - 1:day_last]
        _, acts = self.generate_trajectory(torch.as_tensor(pi0, dtype=torch.float32, device=self.device), 16)
        V, l1, js = ops.backward_value(acts.contiguous(), want_jsd=jsd)
        self.mat_V = V.cpu().numpy()                      # [days, 16, d]
        vals = (js if jsd else l1).cpu().numpy().reshape(-1)
        return float(np.mean(vals)), float(np.std(vals))

    @_with_ctx
    def evaluate_synthetic(self, day_first=1, day_last=26, verbose=0):
        """Mean / std over (day, hour) of sum_ij |P_ij - value_ij| (mfg_synthetic.py:741-812)."""
        m, s = self._evaluate_synthetic(day_first, day_last, False)
        if verbose:
            print('Mean over all hours', m)
            print('Standard deviation', s)
        return m, s

    @_with_ctx
    def evaluate_synthetic_JSD(self, day_first=1, day_last=26, write_file=0, filename='synthetic_log.csv', verbose=0):
        """Mean / std over (day, hour) of sum_i JSD(P_i, value-implied row i) (mfg_synthetic.py:815-899)."""
        m, s = self._evaluate_synthetic(day_first, day_last, True)
        if verbose:
            print('Mean over all hours', m)
            print('Standard deviation', s)
        return m, s
