"""Drop-in ``AC_IRL`` with the call surface of the reference's ``ac_irl.AC_IRL`` (ac_irl.py:31):
max-ent IRL (guided cost learning) around the same actor-critic, batched on the GPU.

The NumPy hot-path methods of the reference class are the mfg_ac2 ones minus calc_reward; here they are
inherited from :class:`discrete_mean_field_game_amd.mfg_ac2.actor_critic` (HIP kernels).  What differs, and
is kept:
  * reward of a transition = reward network r(pi, P) (ac_irl.py:683) -> one batched PyTorch forward over
    [B,d,d] actions per env step; dropout stays active there like in the reference;
  * TD error bootstraps with the running ``discount = gamma**t`` (ac_irl.py:691, :710);
  * episodes are 1-indexed in the learning-rate schedule (ac_irl.py:649, :700, :708);
  * early stop on |theta - prev_theta| < stop_criteria (:726) and the list_policies FIFO (:731);
  * update_reward / reward_iteration / outerloop (ac_irl.py:804-954) with the loss of :390-413, Adam 1e-4.
TensorFlow is replaced by PyTorch-ROCm (networks.RewardNet); ``use_tf`` is accepted and means "build the
reward network".  Disk inputs are optional: pass ``pi0`` / ``pi0_test`` / ``demonstrations`` arrays, or
let the constructor read the reference's directories when they exist.
"""
from __future__ import annotations

import os
import random
import warnings

import numpy as np
import torch

from . import _lib as L
from . import ops
from .mfg_ac2 import EPISODE_STEPS, _with_ctx, actor_critic
from .networks import RewardNet, maxent_irl_loss
from .parallel import all_reduce_gradients_, all_reduce_mean_flat_, broadcast_seed, current_shard, lr_scales
from .reward_learning import RewardTrainer, TrajectoryStore


class AC_IRL(actor_critic):

    def __init__(self, theta=8.64, shift=0, alpha_scale=1e4, d=15, lr_reward=1e-4, num_policies=10, c=2e11,
                 reg='dropout_l1l2', n_fc3=8, n_fc4=4, saved_network=None, use_tf=True, summarize=False, *,
                 pi0=None, pi0_test=None, demonstrations=None, demonstrations_test=None, batch=1, rng='philox',
                 seed=0, update_every='step', precision='mixed', device=None, group=None, verbose=1, check_finite=False):
        super().__init__(theta=theta, shift=shift, alpha_scale=alpha_scale, d=d, pi0=pi0, batch=batch, rng=rng,
                         seed=seed, update_every=update_every, precision=precision, device=device, group=group,
                         verbose=verbose, check_finite=check_finite)
        self.summarize = summarize
        self.theta_initial = theta                      # reset value used by outerloop (ac_irl.py:45, :942)
        self.lr_reward = lr_reward
        self.num_policies = num_policies
        self.c = c
        self.reg = reg
        self.n_fc3 = n_fc3
        self.n_fc4 = n_fc4
        if pi0_test is not None:
            self.mat_pi0_test = np.array(pi0_test, dtype=np.float64)[:, 0:d]
        elif os.path.isdir(os.getcwd() + '/test_normalized_round2'):
            self.init_pi0_test(path_to_dir=os.getcwd() + '/test_normalized_round2', day_start=22)
        else:
            self.mat_pi0_test = self.mat_pi0.copy()
        self.num_start_samples_test = self.mat_pi0_test.shape[0]
        # device-resident demonstrations / D_samp (reward_learning.TrajectoryStore); the list attributes below are views
        self._demo_store = TrajectoryStore(d, EPISODE_STEPS, self.device)
        self._gen_store = TrajectoryStore(d, EPISODE_STEPS, self.device)
        self._demo_list = []
        self._eval_demo_override = None
        self._eval_gen_override = None
        self._trainer = None
        self._stats_host = None
        if demonstrations is not None:
            self.list_demonstrations = demonstrations
        elif os.path.isdir('./actions_2') and os.path.isdir('./train_normalized_round2'):
            self.list_demonstrations = self.read_demonstrations('./train_normalized_round2', './actions_2', 20, 1)
        else:
            self.list_demonstrations = []
        if demonstrations_test is not None:
            self.list_demonstrations_test = demonstrations_test
        elif os.path.isdir('./actions_test_2') and os.path.isdir('./test_normalized_round2'):
            self.list_demonstrations_test = self.read_demonstrations('./test_normalized_round2', './actions_test_2', 20, 22)
        else:
            self.list_demonstrations_test = []
        # list_eval_demo_transitions (ac_irl.py:77) / list_generated = D_samp (:79): properties over the stores, see below
        self.num_demo_samples = 5
        self.num_gen_samples = 5
        self.num_sampled_trajectories = self.num_gen_samples
        self.list_policies = [theta] * self.num_policies
        self.reward_update_count = 0
        self._reward_calls = 0                           # number of reward() calls so far (dropout-mask counter)
        self._reward_sample_offset = 0                   # global index of the first sample of the next reward() call
        self._reward_train_calls = 0                     # update_reward calls so far (dropout-mask key of the training batches)
        self.reward_net = None
        if use_tf:
            self.create_network()
            self.create_training_method()
            if saved_network:
                self.reward_net.load_state_dict(torch.load('saved/' + saved_network, map_location=self.device))

    # ------------------------------------------------------------------ data (next rows, ac_irl.py:164-200)
    def init_pi0_test(self, path_to_dir, day_start=22, verbose=0):
        rows = []
        num_files = len(os.listdir(path_to_dir))
        for num_day in range(day_start, day_start + num_files):
            with open(path_to_dir + '/trend_distribution_day%d.csv' % num_day, 'r') as f:
                first = f.readline()
            rows.append(list(map(float, first.strip().split(' ')))[0:self.d])
        self.mat_pi0_test = np.array(rows, dtype=np.float64)

    def read_demonstrations(self, state_dir, action_dir, dim_action=20, start_day=1):
        """list of trajectories, each a list of 15 (state [d], action [d,d]) pairs (ac_irl.py:164-200).
        State files: 16 rows x >= d; action files: 15 blocks of dim_action rows (blank lines skipped)."""
        num_file_action = len(os.listdir(action_dir))
        out = []
        for idx_day in range(start_day, start_day + num_file_action):
            states = np.loadtxt(state_dir + '/trend_distribution_day%d.csv' % idx_day, delimiter=' ', ndmin=2)
            actions = np.loadtxt(action_dir + '/action_day%d.txt' % idx_day, delimiter=' ', ndmin=2)
            traj = []
            for hour in range(0, 15):
                state = states[hour, 0:self.d]
                action = actions[hour * dim_action:(hour * dim_action + self.d), 0:self.d]
                traj.append((state, action))
            out.append(traj)
        return out

    def get_eval_transitions(self, list_trajectories):
        """One (s,a) per trajectory: index = trajectory index mod 15 (ac_irl.py:203-219)."""
        return [traj[idx % 15] for idx, traj in enumerate(list_trajectories)]

    # ------------------------------------------------------------------ list views of the device stores
    @property
    def list_demonstrations(self):
        """Expert trajectories (ac_irl.py:69).  Assigning a list uploads it once to the device store."""
        return self._demo_list

    @list_demonstrations.setter
    def list_demonstrations(self, trajs):
        self._demo_list = trajs
        # the reference flattens demonstrations of ANY length (ac_irl.py:816-821; only generated trajectories are reshaped to
        # [M, 15]): a list with other lengths stays on the host and update_reward takes the autograd path for it
        self._demo_ragged = any(len(tr) != EPISODE_STEPS for tr in trajs)
        if self._demo_ragged:
            self._demo_store.clear()
        else:
            self._demo_store.assign_list(trajs)
        self._demo_ids = tuple(id(tr) for tr in trajs)
        self._eval_demo_override = None

    @property
    def list_generated(self):
        """D_samp (ac_irl.py:79, :927-932) as the reference's list[n] of list[15] of (pi, P): built from the device store when
        read (cached until D_samp changes).  Assign a whole list to replace D_samp; in-place edits of the returned list are
        not seen by the device store."""
        return self._gen_store.to_list()

    @list_generated.setter
    def list_generated(self, trajs):
        if trajs is not self._gen_store._list or trajs is None:
            self._gen_store.assign_list(trajs if trajs is not None else [])
            self._gen_store._list = trajs if trajs else None       # keep the caller's objects as the list view
        self._gen_ids = tuple(id(tr) for tr in trajs) if trajs else None
        self._eval_gen_override = None

    def _resync_stores(self):
        """In-place edits of a list view do not pass the setter: `ac.list_generated += more`, `.append`, `list[i] = traj`.  A view
        whose LENGTH or whose trajectory OBJECTS (ids: a cheap fingerprint, taken when the list was uploaded) no longer match
        is uploaded again.  (Edits inside a trajectory -- replacing one (pi, P) pair, writing into an array -- are not seen:
        assign the list.)"""
        ids = tuple(id(tr) for tr in self._demo_list)
        if ids != getattr(self, '_demo_ids', ids) or (not getattr(self, '_demo_ragged', False)
                                                      and len(self._demo_list) != len(self._demo_store)):
            self.list_demonstrations = self._demo_list
        view = self._gen_store._list
        if view is not None:
            ids = tuple(id(tr) for tr in view)
            if len(view) != len(self._gen_store) or (getattr(self, '_gen_ids', None) is not None and ids != self._gen_ids):
                self._gen_store.assign_list(view)
                self._gen_store._list = view
                self._gen_ids = ids

    def _is_all_pairs(self, pairs, store, trajs):
        """True if `pairs` is the flattened view [pair for traj in trajs for pair in traj] of the store's current list."""
        n = len(store) * store.steps
        return (trajs is not None and len(pairs) == n and (n == 0 or (pairs[0] is trajs[0][0] and pairs[-1] is trajs[-1][-1])))

    @property
    def list_eval_demo_transitions(self):
        if self._eval_demo_override is not None:
            return self._eval_demo_override
        return [pair for traj in self._demo_list for pair in traj]

    @list_eval_demo_transitions.setter
    def list_eval_demo_transitions(self, pairs):
        self._eval_demo_override = None if self._is_all_pairs(pairs, self._demo_store, self._demo_list) else pairs

    @property
    def list_eval_gen_transitions(self):
        if self._eval_gen_override is not None:
            return self._eval_gen_override
        return [pair for traj in self.list_generated for pair in traj]

    @list_eval_gen_transitions.setter
    def list_eval_gen_transitions(self, pairs):
        self._eval_gen_override = None if self._is_all_pairs(pairs, self._gen_store, self._gen_store._list) else pairs

    # loss / first / second term of the last update_reward (ac_irl.py:846): left on the device by the training kernel, read
    # (one 16-byte copy) only when somebody looks at them
    def _stat(self, k):
        if self._stats_host is None:
            self._stats_host = self._trainer.stats.cpu().numpy().astype(np.float64) if self._trainer is not None else np.zeros(4)
        return float(self._stats_host[k])

    def _set_stat(self, k, value):
        if self._stats_host is None:
            self._stats_host = np.zeros(4)
        self._stats_host[k] = float(value)

    loss_val = property(lambda self: self._stat(0), lambda self, v: self._set_stat(0, v))
    first_term_val = property(lambda self: self._stat(1), lambda self, v: self._set_stat(1, v))
    second_term_val = property(lambda self: self._stat(2), lambda self, v: self._set_stat(2, v))

    # ------------------------------------------------------------------ reward network (ac_irl.py:232-267, :382-427)
    def create_network(self):
        self.reward_net = RewardNet(d=self.d, reg=self.reg, f1=1, k1=5, f2=2, k2=3, n_fc3=self.n_fc3,
                                    n_fc4=self.n_fc4).to(self.device)

    def create_training_method(self):
        """Loss + Adam of ac_irl.py:382-418.  On the GPU (network inside the HIP kernels' range) the whole update is
        mfg_reward_net_train_step on flat parameter / moment buffers (reward_learning.RewardTrainer); `self.optimizer`
        (torch Adam) serves only shapes outside that range."""
        if ops.reward_net_supported(self.reward_net):
            self._trainer = RewardTrainer(self.reward_net, self.lr_reward)
        self.optimizer = torch.optim.Adam(self.reward_net.parameters(), lr=self.lr_reward)

    def _pairs_to_tensors(self, pairs):
        s = torch.as_tensor(np.array([np.asarray(p[0], dtype=np.float32) for p in pairs]), device=self.device)
        a = torch.as_tensor(np.array([np.asarray(p[1], dtype=np.float32) for p in pairs]), device=self.device)
        return s, a

    @_with_ctx
    def reward(self, pi, P):
        """r(pi, P) from the reward network: [B,d], [B,d,d] -> [B] (ac_irl.py:683)."""
        if ops.reward_net_supported(self.reward_net) and pi.is_cuda:
            # fresh dropout masks per call: the FULL call counter goes into the 64-bit Philox key (no wrap-around within
            # a run), the counter's sample index is the GLOBAL one (rank shard offset + n), so ranks draw different masks
            self._reward_calls += 1
            key = ((self.seed + 0x5EED) ^ (self._reward_calls * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
            return ops.reward_net_forward(self.reward_net, pi.contiguous(), P.contiguous(), seed=key,
                                          sample_offset=int(self._reward_sample_offset))
        if not getattr(self, '_warned_reward_fallback', False):                  # shapes outside the kernel's range
            self._warned_reward_fallback = True
            warnings.warn('AC_IRL.reward: the reward network (d=%d, n_fc3=%d, n_fc4=%d) is outside the range of the HIP '
                          'kernel mfg_reward_net_forward (d, n_fc3, n_fc4 <= 32, f2 <= 2, CUDA tensors); evaluating it with the '
                          'PyTorch module instead' % (self.d, self.n_fc3, self.n_fc4), RuntimeWarning, stacklevel=2)
        with torch.no_grad():
            return self.reward_net(pi, P).reshape(-1).float().contiguous()

    @_with_ctx
    def calc_alpha_deriv(self, pi):
        """d alpha / d theta for state pi (ac_irl.py:573-588); stored like the reference."""
        pi_dev, single = self._pi_dev(pi)
        _, ad = ops.alpha(pi_dev, self._theta, self.shift, want_alpha=False)
        ad = ad.cpu().numpy()
        self._alpha_deriv_host = ad[0] if single else ad
        return self._alpha_deriv_host

    # ------------------------------------------------------------------ a9 (IRL flavour)
    @_with_ctx
    def train(self, max_episodes=4000, stop_criteria=0.01, gamma=1, constant=False, lr_critic=0.1, lr_actor=0.001,
              consecutive=100, file_theta='results/theta.csv', file_pi='results/pi.csv',
              file_reward='results/reward.csv', write_file=0, write_all=0, reward_fn=None, *, first_episode=0):
        """Forward actor-critic under the learned reward (ac_irl.py:634-732).  ``reward_fn(pi, P) -> [B]``
        overrides the reward network (used by the parity tests with a closed-form reward).
        first_episode: episodes already run before a resume: `max_episodes` MORE episodes are run, numbered
        first_episode+1 .. first_episode+max_episodes in the lr/(episode+1) schedule (the same meaning as in
        actor_critic.train; the reference always starts at 1)."""
        d, T = self.d, EPISODE_STEPS
        if self.verbose:
            print('----- Starting train -----')
        shard = current_shard(self.batch, self.group)
        if shard.world > self.batch:
            raise ValueError('batch=%d is smaller than the world size %d: every rank needs a trajectory'
                             % (self.batch, shard.world))
        F = ops.num_features(d)
        G = torch.zeros(F + 3, dtype=torch.float64, device=self.device)
        ws = ops.workspace(shard.local_batch, d, self.device)
        dg = (torch.empty(shard.local_batch, dtype=torch.float64, device=self.device),
              torch.empty(shard.local_batch, dtype=torch.float64, device=self.device))
        rfn = reward_fn if reward_fn is not None else self.reward
        Bl = shard.local_batch
        # rollout(T=1) output buffers, reused every step (pi_last is a fresh tensor per step)
        rbufs = {'pi_traj': torch.empty(Bl, 2, d, dtype=torch.float32, device=self.device),
                 'P': torch.empty(Bl, 1, d, d, dtype=torch.float32, device=self.device),
                 'delta': dg[0].view(Bl, 1), 'g': dg[1].view(Bl, 1)}
        fused_episode = (self.update_every == 'rollout' and self.rng == 'philox' and not write_all)
        if fused_episode:
            ws_ep = ops.workspace(Bl * T, d, self.device)
            ebufs = {'pi_traj': torch.empty(Bl, T + 1, d, dtype=torch.float32, device=self.device),
                     'pi_last': torch.empty(Bl, d, dtype=torch.float32, device=self.device),
                     'P': torch.empty(Bl, T, d, d, dtype=torch.float32, device=self.device),
                     'reward': torch.empty(Bl * T, dtype=torch.float32, device=self.device),
                     'delta': torch.empty(Bl, T, dtype=torch.float64, device=self.device),
                     'g': torch.empty(Bl, T, dtype=torch.float64, device=self.device)}
        # ... and with the reward network's HIP kernel on one GPU the whole episode is ONE native call (mfg_train_rollout_irl:
        # start states drawn inside the rollout kernel, the network reads its states in place from pi_traj, sums + update)
        native_rollout = (fused_episode and shard.world == 1 and reward_fn is None and self.trace is None
                          and self.reward_net is not None and ops.reward_net_supported(self.reward_net)
                          and self._device_draw())
        # per-step updates on one GPU with the reward network's HIP kernel: the whole episode (15 x [sample + transition +
        # score | reward net | batch sums + update]) is issued by native code (mfg_train_episode_irl)
        native_episode = (self.update_every == 'step' and self.rng == 'philox' and shard.world == 1 and reward_fn is None
                          and self.trace is None and not write_all and self.reward_net is not None
                          and ops.reward_net_supported(self.reward_net))
        if native_episode:
            nbufs = dict(ops.episode_buffers(Bl, d, self.device), P=rbufs['P'], pi=torch.empty(Bl, d, dtype=torch.float32, device=self.device))
        # per-episode return accumulators of the native paths: ONE zeroed buffer per train() call instead of a fill kernel per episode
        ep_acc = torch.zeros(max_episodes + 1, dtype=torch.float64, device=self.device) if (native_rollout or native_episode) else None
        prev_theta = float(self._theta.cpu()[0])
        list_reward = []
        episode = 0
        pi = None
        device_draw = self._device_draw()       # batched Philox runs: start states drawn on the device (mfg_draw_start)
        for episode in range(1 + first_episode, first_episode + max_episodes + 1):
            sc, sa = lr_scales(episode, constant)          # lr/(episode+1) with the 1-indexed episode (:700)
            if native_rollout:
                self._reward_calls += 1                    # the keys of ONE reward() call over the [B*T] transitions
                key = ((self.seed + 0x5EED) ^ (self._reward_calls * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
                total_reward = ep_acc[episode - first_episode:episode - first_episode + 1]
                ops.train_rollout_irl(self._mat_pi0_dev, None, T, self._theta, self.shift, self.alpha_scale, self._w, gamma,
                                      lr_critic * sc, lr_actor * sa, self.reward_net, G, ws_ep, ebufs, seed=self.seed,
                                      first_step=self._rng_step, traj_offset=shard.traj_offset, rn_key=key,
                                      rn_sample_offset=shard.traj_offset * T, reward_acc=total_reward, precision=self.precision)
                self._rng_step += T
                self._reward_sample_offset = shard.traj_offset * T
                self._theta_is_array = True
                pi = ebufs['pi_last']
                list_reward.append(total_reward)           # (mean reward per transition; scaled by T where it is reported)
                if self.check_finite:
                    self._raise_if_not_finite(pi, episode)
                if episode % consecutive == 0:
                    self._report_irl(list_reward, consecutive, pi, write_file, file_theta, file_pi, file_reward, scale=float(T))
                    list_reward = []
                if stop_criteria != -1:
                    cur = float(self._theta.cpu()[0])
                    if abs(cur - prev_theta) < stop_criteria:
                        break
                    prev_theta = cur
                continue
            if native_episode and device_draw:
                pi = nbufs['pi']                           # (output only: the start states are drawn inside the native call)
            elif device_draw:
                _, pi = ops.draw_start(self._mat_pi0_dev, Bl, self.seed, self._rng_step, shard.traj_offset)
            else:
                pi = ops.gather_start(self._mat_pi0_dev, self._draw_start(shard))       # ac_irl.py:655
            discount = 1.0
            total_reward = (ep_acc[episode - first_episode:episode - first_episode + 1] if native_episode
                            else torch.zeros(1, dtype=torch.float64, device=self.device))
            if fused_episode:
                # one update per episode: theta and w are fixed over the 15 steps, so the whole episode is THREE launches:
                # the fused rollout (running discount gamma^t, P of every step materialised for the network), one
                # reward-net pass over all B*T transitions, one gradient pass that folds the rewards into delta
                o = ops.rollout(pi, T, self._theta, self.shift, self.alpha_scale, w=self._w, gamma=gamma,
                                reward_kind=L.REWARD_EXTERNAL, seed=self.seed, first_step=self._rng_step,
                                traj_offset=shard.traj_offset, td=True, write_P=True, discount_pow=True,
                                precision=self.precision, out=ebufs)
                self._rng_step += T
                states = o['pi_traj'][:, :T].reshape(Bl * T, d)
                self._reward_sample_offset = shard.traj_offset * T
                r = rfn(states, o['P'].view(Bl * T, d, d))
                if shard.world == 1:
                    ops.grad_apply(o['pi_traj'], o['delta'].view(-1), o['g'].view(-1), r, G, ws_ep, lr_critic * sc,
                                   lr_actor * sa, self._w, self._theta, total_reward, T=T, add_reward=True)
                else:
                    ops.grad_accumulate(o['pi_traj'], o['delta'].view(-1), o['g'].view(-1), r, G, ws_ep, T=T, add_reward=True)
                    all_reduce_gradients_(G, self.group)
                    ops.apply_update(G, d, lr_critic * sc, lr_actor * sa, self._w, self._theta, total_reward)
                total_reward = total_reward * T
                self._theta_is_array = True
                pi = o['pi_last']
            if native_episode:
                ops.train_episode_irl(pi, T, self._theta, self.shift, self.alpha_scale, self._w, gamma, lr_critic * sc,
                                      lr_actor * sa, self.reward_net, G, ws, nbufs, seed=self.seed, first_step=self._rng_step,
                                      traj_offset=shard.traj_offset, rn_seed=self.seed + 0x5EED, rn_call0=self._reward_calls,
                                      rn_sample_offset=shard.traj_offset, reward_acc=total_reward, precision=self.precision,
                                      mat_pi0=self._mat_pi0_dev if device_draw else None)
                self._rng_step += T
                self._reward_calls += T                    # the dropout-mask keys of T reward() calls were consumed
                self._reward_sample_offset = shard.traj_offset
                self._theta_is_array = True
            for step in range(0 if (fused_episode or native_episode) else T):
                acc = (self.update_every == 'rollout' and step > 0)
                step_applied = False
                if self.rng == 'philox':
                    # ONE launch samples P, takes the transition and evaluates everything of the TD step that does not
                    # need the reward (score g, gamma V(pi') - V(pi)); P is materialised for the reward network, whose
                    # output is folded in by the gradient kernel (delta += r) -- no second pass over P
                    o = ops.rollout(pi, 1, self._theta, self.shift, self.alpha_scale, w=self._w, gamma=discount,
                                    reward_kind=L.REWARD_EXTERNAL, seed=self.seed, first_step=self._rng_step,
                                    traj_offset=shard.traj_offset, td=True, write_P=True, precision=self.precision,
                                    out=rbufs)
                    self._rng_step += 1
                    P = o['P'].view(shard.local_batch, d, d)
                    pi_next = o['pi_last']
                    if write_all:
                        self._write_all(pi, P, step + 1)
                    self._reward_sample_offset = shard.traj_offset
                    r = rfn(pi, P)
                    if self.update_every == 'step' and shard.world == 1:
                        # sums + update in one launch (the step is core -> reward net -> this)
                        ops.grad_apply(pi, dg[0], dg[1], r, G, ws, lr_critic * sc, lr_actor * sa, self._w, self._theta,
                                       total_reward, add_reward=True)
                        step_applied = True
                    else:
                        ops.grad_accumulate(pi, dg[0], dg[1], r, G, ws, add_reward=True, accumulate=acc)
                else:
                    P = self._sample(pi, shard.traj_offset, snapshot=False)
                    pi_next, _ = ops.step_given_P(pi, P, want_reward=False)
                    if write_all:
                        self._write_all(pi, P, step + 1)
                    self._reward_sample_offset = shard.traj_offset
                    r = rfn(pi, P)
                    ops.td_pg_accumulate(pi, pi_next, P, r, self._w, self._theta, self.shift, discount, G=G, ws=ws,
                                         precision=self.precision, out=dg, accumulate=acc)
                if self.update_every == 'step':
                    if not step_applied:
                        all_reduce_gradients_(G, self.group)
                        ops.apply_update(G, d, lr_critic * sc, lr_actor * sa, self._w, self._theta, total_reward)
                    self._theta_is_array = True
                    if self.trace is not None:
                        self.trace.append(float(self._theta.cpu()[0]))
                discount = discount * gamma
                pi = pi_next
            if self.update_every == 'rollout' and not fused_episode:
                all_reduce_gradients_(G, self.group)
                ops.apply_update(G, d, lr_critic * sc, lr_actor * sa, self._w, self._theta, total_reward)
                total_reward = total_reward * T
                self._theta_is_array = True
            list_reward.append(total_reward)
            if self.check_finite:
                self._raise_if_not_finite(pi, episode)
            if episode % consecutive == 0:
                self._report_irl(list_reward, consecutive, pi, write_file, file_theta, file_pi, file_reward)
                list_reward = []
            if stop_criteria != -1:
                cur = float(self._theta.cpu()[0])
                if abs(cur - prev_theta) < stop_criteria:
                    break
                prev_theta = cur
        self.list_policies = (self.list_policies + [self.theta])[1:]            # record this policy (:731)
        self.episodes_run = episode
        self._check_status()
        if self.verbose:
            print('----- Exiting train at episode %d with theta %f -----' % (episode, float(np.ravel(self.theta)[0])))

    def _report_irl(self, list_reward, consecutive, pi, write_file, file_theta, file_pi, file_reward, scale=1.0):
        """The `consecutive`-episode report of train() (ac_irl.py:714-724): average episode return, theta, one final state."""
        reward_avg = float(torch.cat(list_reward).sum().cpu()) * scale / consecutive
        pi_host = pi[0].cpu().numpy().astype(np.float64)
        if self.verbose:
            print('Theta\n', self.theta)
            print('pi\n', pi_host)
            print('Average reward during previous %d episodes: ' % consecutive, str(reward_avg))
        if write_file:
            self.train_log(np.ravel(self.theta), file_theta, '%.5e')
            self.train_log(pi_host, file_pi, '%.3e')
            self.train_log(np.array([reward_avg]), file_reward, '%.3e')

    # ------------------------------------------------------------------ a10
    @_with_ctx
    def generate_trajectories(self, n, from_test=False):
        """n trajectories of 15 (state, action) pairs under the current policy (ac_irl.py:735-767)."""
        mat_dev = self._mat_pi0_dev
        num = self.num_start_samples
        if from_test:
            mat_dev = torch.as_tensor(np.ascontiguousarray(self.mat_pi0_test, dtype=np.float32), device=self.device)
            num = self.num_start_samples_test
        T = 15
        if self.rng == 'numpy':
            out = []
            for _ in range(n):                                 # the reference's interleaved RNG order
                idx = torch.as_tensor(np.array([np.random.randint(num)], dtype=np.int32), device=self.device)
                pi = ops.gather_start(mat_dev, idx)
                traj = []
                for _h in range(T):
                    P = self._sample(pi)
                    traj.append((pi[0].cpu().numpy().astype(np.float64), P[0].cpu().numpy().astype(np.float64)))
                    pi, _r = ops.step_given_P(pi, P, want_reward=False)
                out.append(traj)
            return out
        pis, Ps = self._generate_device(n, mat_dev)
        pis = pis.cpu().numpy().astype(np.float64)
        Ps = Ps.cpu().numpy().astype(np.float64)
        return [[(pis[b, t], Ps[b, t]) for t in range(T)] for b in range(n)]

    def _generate_device(self, n, mat_dev=None):
        """generate_trajectories without the host copy: (pi_traj [n,16,d], P [n,15,d,d]) device tensors straight from
        mfg_rollout(WRITE_P).  Start states drawn on the device like train()'s (mfg_draw_start keyed by seed / step /
        trajectory id): every rank of a multi-GPU job generates the identical D_samp without sharing a host RNG stream."""
        mat_dev = self._mat_pi0_dev if mat_dev is None else mat_dev
        T = EPISODE_STEPS
        off = self._gen_offset(n)
        _, pi0 = ops.draw_start(mat_dev, n, self.seed, self._rng_step, off)
        r = ops.rollout(pi0, T, self._theta, self.shift, self.alpha_scale, seed=self.seed, first_step=self._rng_step,
                        traj_offset=off, td=False, write_P=True, precision=self.precision)
        self._rng_step += T
        return r['pi_traj'], r['P']

    def _gen_offset(self, n):
        off = getattr(self, '_gen_traj_counter', 1 << 40)       # disjoint from the training trajectory ids
        self._gen_traj_counter = off + n
        return off

    # ------------------------------------------------------------------ checkpoint / resume
    @_with_ctx
    def state_dict(self):
        """actor_critic.state_dict() plus the reward network, its optimiser, the policy FIFO and the counters of the
        outer loop (the reference only saves the TF reward net, ac_irl.py:948)."""
        st = super().state_dict()
        ver, internal, gauss = random.getstate()
        if len(self._gen_store):
            gs, ga = self._gen_store.gather()
            gen_pi, gen_P = gs.cpu().double(), ga.cpu().double()
        else:
            gen_pi = gen_P = None
        st.update({'reward_net': self.reward_net.state_dict(), 'optimizer': self.optimizer.state_dict(),
                   'list_policies': [float(np.ravel(t)[0]) for t in self.list_policies],
                   'theta_initial': float(np.ravel(self.theta_initial)[0]),
                   'reward_update_count': int(getattr(self, 'reward_update_count', 0)),
                   'reward_calls': int(self._reward_calls),
                   'gen_traj_counter': int(getattr(self, '_gen_traj_counter', 1 << 40)),
                   # D_samp (ac_irl.py:79, :927-932) as two tensors [M,15,d], [M,15,d,d]
                   'list_generated_pi': gen_pi if gen_pi is not None else torch.zeros(0),
                   'list_generated_P': gen_P if gen_P is not None else torch.zeros(0),
                   'reward_trainer': self._trainer.state_dict() if self._trainer is not None else {},
                   'reward_train_calls': int(self._reward_train_calls),
                   # host RNG streams a resumed run consumes: Python `random` (update_reward's random.sample) and
                   # torch's CPU / device generators (dropout in the training-mode reward net)
                   'py_random_version': int(ver), 'py_random_state': torch.tensor(internal, dtype=torch.int64),
                   'py_random_gauss': gauss,
                   'torch_rng_state': torch.get_rng_state(),
                   'torch_cuda_rng_state': torch.cuda.get_rng_state(self.device)})
        return st

    @_with_ctx
    def load_state_dict(self, state, restore_np_random=True):
        super().load_state_dict(state, restore_np_random)
        self.reward_net.load_state_dict(state['reward_net'])
        self.optimizer.load_state_dict(state['optimizer'])
        self.list_policies = list(state['list_policies'])
        self.reward_update_count = int(state['reward_update_count'])
        self._gen_traj_counter = int(state['gen_traj_counter'])
        self.theta_initial = state.get('theta_initial', self.theta_initial)
        self._reward_calls = int(state.get('reward_calls', 0))
        if self._trainer is not None:
            if state.get('reward_trainer'):
                self._trainer.load_state_dict(state['reward_trainer'])
            else:
                # a checkpoint from before the HIP training step (round <= 4): its Adam state lives in the torch optimiser
                # only -- take the moments and the step count from there (same parameter order as the flat buffer)
                self._trainer.seed_from_torch_adam(self.optimizer, list(self.reward_net.parameters()))
        self._reward_train_calls = int(state.get('reward_train_calls', 0))
        self._stats_host = None
        if 'list_generated_pi' in state:
            self._gen_store.clear()
            if state['list_generated_pi'].numel():
                self._gen_store.push(state['list_generated_pi'], state['list_generated_P'])
            self._eval_gen_override = None                       # all of D_samp, as outerloop keeps it (:934)
        if restore_np_random and 'py_random_state' in state:
            random.setstate((int(state['py_random_version']), tuple(int(v) for v in state['py_random_state']),
                             state['py_random_gauss']))
            torch.set_rng_state(state['torch_rng_state'].cpu())
            torch.cuda.set_rng_state(state['torch_cuda_rng_state'].cpu(), self.device)

    # ------------------------------------------------------------------ importance weights (ac_irl.py:270-379)
    @_with_ctx
    def calc_pdf_action(self, theta, action, state, log=False):
        """q(a_t; s_t, theta) of the product-Dirichlet policy (ac_irl.py:270-289); `log=True` returns ln q (the
        density itself under/overflows fp64 long before d=15 rows are multiplied)."""
        s, a = self._pairs_to_tensors([(state, action)])
        th = torch.tensor([float(np.ravel(theta)[0])], dtype=torch.float64, device=self.device)
        lq = float(ops.policy_logpdf(s, a, th, self.shift)[0, 0].cpu())
        return lq if log else float(np.exp(lq))

    @_with_ctx
    def calc_z(self, list_trajectories, log=False):
        """z(traj_j) = [1/k sum_k q_k(traj_j)]^-1 over the policies in `list_policies` (ac_irl.py:292-321, :324-379;
        alpha lower-bounded by 1+1e-6 like :359).  One HIP launch for all (transition, policy) pairs; the products
        over topics / time and the sum over policies are done in log space instead of the reference's
        divide-by-`c` normaliser.  Returns z [M] (or ln z with `log=True`)."""
        thetas = torch.as_tensor(np.array([float(np.ravel(t)[0]) for t in self.list_policies], dtype=np.float64),
                                 device=self.device)
        M, T = len(list_trajectories), len(list_trajectories[0])
        s, a = self._pairs_to_tensors([pair for traj in list_trajectories for pair in traj])
        lq = ops.policy_logpdf(s, a, thetas, self.shift, 1.0, 1.0 + 1e-6).view(M, T, -1).sum(1)
        lq = lq - float(np.log(self.num_start_samples))
        lz = float(np.log(thetas.numel())) - torch.logsumexp(lq, dim=1)
        out = lz if log else torch.exp(lz)
        return out.cpu().numpy()

    # ------------------------------------------------------------------ reward learning (ac_irl.py:804-897)
    @_with_ctx
    def update_reward(self, summary=False, iteration=0):
        """One gradient step on the reward network (ac_irl.py:804-846).  The batch is drawn with the reference's
        `random.sample` calls -- on INDEX ranges: random.sample(population, k) picks positions from len(population) alone, so
        the same trajectories are chosen and the host stream advances identically -- and the update itself is
        mfg_reward_net_train_step on the device stores: two launches, nothing copied, no synchronisation."""
        self._resync_stores()
        ragged = getattr(self, '_demo_ragged', False)
        nd_all, ng_all = (len(self._demo_list) if ragged else len(self._demo_store)), len(self._gen_store)
        if nd_all >= self.num_demo_samples:
            demo_idx = random.sample(range(nd_all), self.num_demo_samples)
        else:
            demo_idx = list(range(nd_all))
        if ng_all >= self.num_gen_samples:
            gen_idx = random.sample(range(ng_all), self.num_gen_samples)
        else:
            gen_idx = list(range(ng_all))
        self._reward_train_calls += 1
        n_tr = (len(demo_idx) + len(gen_idx)) * EPISODE_STEPS
        fits = n_tr <= 2048 and n_tr * (1 + self.n_fc3) * 4 <= 60 * 1024            # limits of mfg_reward_net_train_step's combine kernel
        if (self._trainer is not None and fits and not ragged and len(demo_idx) <= L.RN_TRAIN_MAX_TRAJ
                and len(gen_idx) <= L.RN_TRAIN_MAX_TRAJ):
            key = ((self.seed + 0x7EA1) ^ (self._reward_train_calls * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
            dist = torch.distributed
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
            self._trainer.step(self._demo_store, [self._demo_store.rows[i] for i in demo_idx], self._gen_store,
                               [self._gen_store.rows[i] for i in gen_idx], self.num_demo_samples, key, grad_only=multi)
            if multi:
                # replicated reward network: ONE all-reduce of the flat gradient, averaged; the ranks train on
                # identical batches (_sync_host_sampler), so this only keeps them in lock-step
                all_reduce_mean_flat_([self._trainer.grad], self.group)
                self._trainer.apply_grad()
            self._stats_host = None
            return
        self._update_reward_torch(demo_idx, gen_idx)

    def _update_reward_torch(self, demo_idx, gen_idx):
        """update_reward through PyTorch autograd: only for reward networks outside the range of the HIP kernels (d > 32, ...)."""
        demos, gens = self._demo_list, self.list_generated
        ds, da = self._pairs_to_tensors([pair for i in demo_idx for pair in demos[i]])
        gs, ga = self._pairs_to_tensors([pair for i in gen_idx for pair in gens[i]])
        self.reward_net.train()
        r_demo = self.reward_net(ds, da)
        r_gen = self.reward_net(gs, ga)
        reg = self.reward_net.regularization() if self.reward_net.use_l1l2 else None
        loss, first, second = maxent_irl_loss(r_demo, r_gen, self.num_demo_samples, len(gen_idx), reg)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self._all_reduce_reward_grads()
        if self._trainer is not None:
            # ONE optimiser state: a batch outside the HIP training step's range (more than 2 048 transitions, ragged
            # demonstrations) still steps the trainer's Adam moments and step count (mfg_reward_net_adam, tf.train.AdamOptimizer's
            # formula) -- the torch optimiser would keep moments and a bias-correction step of its own
            tr = self._trainer
            off = 0
            for prm in self.reward_net.parameters():
                n = prm.numel()
                if prm.grad is None:
                    tr.grad[off:off + n].zero_()
                else:
                    tr.grad[off:off + n].copy_(prm.grad.reshape(-1))
                off += n
            tr.apply_grad()
        else:
            self.optimizer.step()
        self.loss_val = float(loss.detach().cpu())
        self.first_term_val = float(first.detach().cpu())
        self.second_term_val = float(second.detach().cpu())

    def _all_reduce_reward_grads(self):
        """Replicated reward network: ONE all-reduce of the flattened gradient, averaged over the ranks.  The
        ranks train on identical batches (see _sync_host_sampler), so this only keeps them in lock-step against rounding
        differences."""
        all_reduce_mean_flat_([p.grad for p in self.reward_net.parameters() if p.grad is not None], self.group)

    def _sync_host_sampler(self):
        """update_reward draws its demonstration / generated trajectories with Python's `random` (ac_irl.py:814-829) and
        the training-mode reward net draws dropout masks from torch's generator.  Several ranks must pick the SAME batches
        and masks: rank 0 draws one value and broadcasts it, every rank re-seeds `random` and torch from it.  Once per
        reward_iteration, not per update; a single process keeps its streams untouched."""
        dist = torch.distributed
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        s = broadcast_seed(random.getrandbits(62), self.group, self.device)
        random.seed(s)
        torch.manual_seed(s)

    def _eval_transitions(self, override, store):
        """(states [N,d], actions [N,d,d]) device tensors of an evaluation set: the whole store unless the caller assigned
        its own list of pairs (e.g. get_eval_transitions)."""
        if override is not None:
            return self._pairs_to_tensors(override)
        if store is self._demo_store and getattr(self, '_demo_ragged', False):       # host-resident demonstrations
            pairs = [pair for traj in self._demo_list for pair in traj]
            if pairs:
                return self._pairs_to_tensors(pairs)
        return store.gather_flat()

    def _eval_reward_averages(self):
        """Mean reward over the demonstration / generated evaluation transitions (ac_irl.py:868-883): two forward launches
        and ONE host read for both numbers."""
        self._resync_stores()
        ds, da = self._eval_transitions(self._eval_demo_override, self._demo_store)
        gs, ga = self._eval_transitions(self._eval_gen_override, self._gen_store)
        self._reward_sample_offset = 0
        with torch.no_grad():
            rd = self.reward(ds, da) if ds.shape[0] else torch.zeros(1, device=self.device)
            rg = self.reward(gs, ga) if gs.shape[0] else torch.zeros(1, device=self.device)
            sums = torch.stack([rd.double().sum(), rg.double().sum()]).cpu().numpy()
        nd, ng = ds.shape[0], gs.shape[0]
        return (float(sums[0]) / nd if nd else float('nan')), (float(sums[1]) / ng if ng else float('nan'))

    @_with_ctx
    def reward_iteration(self, max_iterations=500, stop_criteria=0.01, iter_check=10):
        prev_reward_demo_avg = -100
        self._sync_host_sampler()
        if self.verbose:
            print('----- Starting reward_iteration -----')
        it = 0
        for it in range(1, max_iterations + 1):
            self.reward_update_count += 1
            if it % iter_check != 0:
                self.update_reward(summary=False)
                continue
            self.update_reward(summary=False, iteration=self.reward_update_count)
            reward_demo_avg, reward_gen_avg = self._eval_reward_averages()
            if self.verbose:
                print('Reward iteration %d' % it)
                print('Reward demo avg %f | Reward gen avg %f' % (reward_demo_avg, reward_gen_avg))
                print('First %f | Second %f | Loss %f' % (self.first_term_val, self.second_term_val, self.loss_val))
            if np.isnan(reward_demo_avg) or np.isnan(reward_gen_avg):
                break
            if os.path.isdir('results'):
                with open('results/reward_training.csv', 'a') as f:
                    f.write('%f,%f\n' % (reward_demo_avg, reward_gen_avg))
            if stop_criteria != -1 and abs(reward_demo_avg - prev_reward_demo_avg) < stop_criteria:
                break
            prev_reward_demo_avg = reward_demo_avg
        if self.verbose:
            print('----- Exiting reward_iteration at iter %d -----' % it)

    # ------------------------------------------------------------------ experiment harnesses of the reference (ac_irl.py:961-1050)
    @_with_ctx
    def test_convergence(self, num_iterations=500, num_gen_from_policy=5, iter_check=10, filename='reward_convergence.csv'):
        """Train the reward network on demonstrations and a FIXED generated set from the current (fixed) policy and log the average
        rewards every `iter_check` updates (ac_irl.py:961-1005): D_samp is generated once on the device, the updates are
        mfg_reward_net_train_step, the averages two forward launches.  Writes results/<filename> when `results/` exists;
        returns the logged rows [(iteration, reward_demo_avg, reward_gen_avg), ...]."""
        if self.rng == 'philox':
            self._gen_store.clear()
            self._gen_store.push(*self._generate_device(num_gen_from_policy * self.num_policies))
        else:
            self.list_generated = self.generate_trajectories(num_gen_from_policy * self.num_policies)
        self._eval_gen_override = None
        write = os.path.isdir('results')
        if write:
            with open('results/' + filename, 'w') as f:
                f.write('iteration,reward_demo_avg,reward_gen_avg\n')
        self._sync_host_sampler()
        rows = []
        for it in range(1, num_iterations + 1):
            if it % iter_check != 0:
                self.update_reward(summary=False)
                continue
            self.update_reward(summary=False, iteration=it)
            reward_demo_avg, reward_gen_avg = self._eval_reward_averages()
            if self.verbose:
                print('Iteration %d' % it)
                print('Reward demo avg %f | Reward gen avg %f' % (reward_demo_avg, reward_gen_avg))
                print('First %f | Second %f | Loss %f' % (self.first_term_val, self.second_term_val, self.loss_val))
            rows.append((it, reward_demo_avg, reward_gen_avg))
            if write:
                with open('results/' + filename, 'a') as f:
                    f.write('%d,%f,%f\n' % (it, reward_demo_avg, reward_gen_avg))
            if np.isnan(reward_demo_avg) or np.isnan(reward_gen_avg):
                break
        return rows

    @_with_ctx
    def test_reward_network(self):
        """Average reward of the fixed network over the training demonstrations, the test demonstrations and as many freshly
        generated trajectories as there are training demonstrations (ac_irl.py:1008-1046).  Returns the reference's tuple
        (reward_demo_avg_train, reward_demo_avg_test, reward_gen_avg); a missing test set gives nan."""
        num_demos = len(self._demo_store)
        if self.rng == 'philox':
            self._gen_store.clear()
            if num_demos:
                self._gen_store.push(*self._generate_device(num_demos))
        else:
            self.list_generated = self.generate_trajectories(num_demos)
        self._eval_gen_override = None
        self._eval_demo_override = None
        train_avg, gen_avg = self._eval_reward_averages()
        test = [pair for traj in self.list_demonstrations_test for pair in traj]
        if test:
            ts, ta = self._pairs_to_tensors(test)
            self._reward_sample_offset = 0
            with torch.no_grad():
                test_avg = float(self.reward(ts, ta).double().sum().cpu()) / len(test)
        else:
            test_avg = float('nan')
        if self.verbose:
            print('Avg reward demo train %f | Avg reward demo test %f | Avg reward gen %f' % (train_avg, test_avg, gen_avg))
        return train_avg, test_avg, gen_avg

    @_with_ctx
    def evaluate(self, theta=8.86349, shift=0.5, alpha_scale=1e4, d=15, episode_length=16, indir='test_normalized_round2',
                 outfile='eval_mfg_round2/validation.csv', write_header=0):
        """actor_critic.evaluate with the defaults of the reference's AC_IRL.evaluate (ac_irl.py:1495: d = 15, validation.csv)."""
        return super().evaluate(theta=theta, shift=shift, alpha_scale=alpha_scale, d=d, episode_length=episode_length, indir=indir,
                                outfile=outfile, write_header=write_header)

    @_with_ctx
    def outerloop(self, num_iterations=20, num_gen_from_policy=5, max_reward_iterations=100,
                  max_forward_episodes=200, gamma=1, constant=False, lr_critic=0.1, lr_actor=0.001, *, first_iteration=0,
                  final_training=True):
        """Alternate reward updates and forward solves (ac_irl.py:900-954); returns the final theta.
        Checkpoint / resume (the reference cannot): `final_training=False` stops after iteration `num_iterations - 1`
        without the closing 2000-episode forward solve, so that `state_dict()` can be saved there; after
        `load_state_dict`, `first_iteration=k` continues at iteration k -- D_samp (list_generated), the reward-update counter
        and the CSV log of the checkpointed run are kept.  outerloop(n) == outerloop(k, final_training=False) ->
        save / load -> outerloop(n, first_iteration=k), bit for bit."""
        write = 1 if os.path.isdir('results') else 0
        on_device = self.rng == 'philox'                    # D_samp filled straight from mfg_rollout(WRITE_P), no host copy
        if first_iteration == 0:
            if on_device:
                self._gen_store.clear()
                self._gen_store.push(*self._generate_device(num_gen_from_policy * self.num_policies))
            else:
                self.list_generated = self.generate_trajectories(num_gen_from_policy * self.num_policies)
            self.reward_update_count = 0
            if write:
                with open('results/reward_training.csv', 'w') as f:
                    f.write('reward_demo_avg,reward_gen_avg\n')
        for it in range(first_iteration, num_iterations):
            if self.verbose:
                print('########## Outerloop iteration %d ##########' % it)
            if on_device:
                self._gen_store.push(*self._generate_device(num_gen_from_policy), drop=num_gen_from_policy)
            else:
                list_generated = self.generate_trajectories(num_gen_from_policy)
                self.list_generated = (self.list_generated + list_generated)[num_gen_from_policy:]
            self._eval_gen_override = None                   # list_eval_gen_transitions = every pair of D_samp (:934)
            self.reward_iteration(max_iterations=max_reward_iterations, stop_criteria=0.0001, iter_check=10)
            self.theta = self.theta_initial
            self.train(max_forward_episodes, -1, gamma, constant, lr_critic, lr_actor, consecutive=100,
                       write_file=write, write_all=0)
        if not final_training:
            return self.theta
        if os.path.isdir('log'):
            torch.save(self.reward_net.state_dict(), 'log/model_%s_%d_%d.ckpt' % (self.reg, self.n_fc3, self.n_fc4))
        if self.verbose:
            print('********** Final forward training **********')
        self.theta = self.theta_initial
        self.train(2000, -1, gamma, constant, lr_critic, lr_actor, consecutive=100, write_file=write, write_all=0)
        return self.theta
