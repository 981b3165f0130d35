"""Thin tensor-level wrappers over the C ABI: torch owns device memory and streams, HIP does the math.

Every function takes CUDA(ROCm) tensors, enqueues on torch's current stream and returns tensors.
Shapes follow include/mfg_hip.h: pi [B,d] fp32, P [B,d,d] fp32, theta / w fp64 device tensors.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk_f32(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError('%s must be a contiguous float32 CUDA tensor' % name)
    return t


def _chk_f64(t, name):
    if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise ValueError('%s must be a contiguous float64 CUDA tensor' % name)
    return t


def _ptr(t):
    """Device address of a tensor; None -> NULL; a plain int is taken as an address already (e.g. one entry of a
    per-episode accumulator array: base.data_ptr() + 8 * k, without building a tensor view per episode)."""
    if t is None or isinstance(t, int):
        return t
    return t.data_ptr()


def init():
    """One-time per-device setup of the HIP library (optional; done lazily otherwise)."""
    L.check(L.lib().mfg_init(), 'mfg_init')


def status(synchronize: bool = True) -> int:
    """Bits of the device status word (0 = healthy; _lib.STATUS_MIXED_RANGE: a mixed-precision sampling launch found
    theta outside the range of its separable exponential, include/mfg_hip.h).  synchronize=True waits for the current
    stream first so that every launch issued so far has reported."""
    import ctypes as C
    if synchronize:
        torch.cuda.current_stream().synchronize()
    bits = C.c_uint(0)
    L.lib().mfg_status(C.byref(bits))
    return int(bits.value)


def clear_status():
    L.check(L.lib().mfg_clear_status(), 'mfg_clear_status')


class Context:
    """An mfg_ctx_t (include/mfg_hip.h): the mutable library state of ONE model instance -- its status word.  `bind()` makes it
    the calling thread's current context: every ops.* call from this thread then reports into / is refused on this context's
    word, not another instance's.  The drop-in classes create one each and bind it at the start of every public method."""

    def __init__(self, device=None):
        import ctypes as C
        self._ptr = None
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        out = C.c_void_p()
        with torch.cuda.device(dev):
            L.check(L.lib().mfg_ctx_create(C.byref(out)), 'mfg_ctx_create')
        self._ptr = out.value
        self.device = dev

    def bind(self):
        lib = L.lib()
        if lib.mfg_ctx_current() != self._ptr:
            with torch.cuda.device(self.device):
                L.check(lib.mfg_ctx_bind(self._ptr), 'mfg_ctx_bind')
        return self

    @staticmethod
    def unbind():
        L.check(L.lib().mfg_ctx_bind(None), 'mfg_ctx_bind')

    KEEP = object()        # bind_scoped(): this context was bound already, restore() has nothing to do

    def bind_scoped(self):
        """bind() that returns what restore() needs to put the calling thread's previous binding back: KEEP if this context
        was current already (nested public methods: one ctypes call), else the previous context's address (None = the
        device's default context)."""
        prev = L.lib().mfg_ctx_current()
        if prev == self._ptr:
            return Context.KEEP
        with torch.cuda.device(self.device):
            L.check(L.lib().mfg_ctx_bind(self._ptr), 'mfg_ctx_bind')
        return prev

    @staticmethod
    def restore(prev):
        if prev is Context.KEEP:
            return
        lib = L.lib()
        # (the previous context may have been destroyed meanwhile, or belong to another device than the current one: then the
        #  thread goes back to the default context -- never to a stale pointer)
        if prev is None or lib.mfg_ctx_bind(prev) != 0:
            lib.mfg_ctx_bind(None)

    def status(self, synchronize=True) -> int:
        import ctypes as C
        if synchronize:
            torch.cuda.current_stream(self.device).synchronize()
        bits = C.c_uint(0)
        L.lib().mfg_ctx_status(self._ptr, C.byref(bits))
        return int(bits.value)

    def clear_status(self):
        L.check(L.lib().mfg_ctx_clear_status(self._ptr), 'mfg_ctx_clear_status')

    def close(self):
        if self._ptr is not None:
            try:
                L.lib().mfg_ctx_destroy(self._ptr)           # (also unbinds it from THIS thread)
            finally:
                self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def num_features(d: int) -> int:
    return int(L.lib().mfg_num_features(d))


def feature_index(i: int, j: int, d: int) -> int:
    return int(L.lib().mfg_feature_index(i, j, d))


def workspace(N: int, d: int, device) -> torch.Tensor:
    n = int(L.lib().mfg_workspace_bytes(N, d))
    return torch.zeros(max(n, 8) // 8, dtype=torch.float64, device=device)   # zeroed: trailing control block (mfg_hip.h)


def gather_start(mat_pi0, idx):
    _chk_f32(mat_pi0, 'mat_pi0')
    if idx.dtype != torch.int32 or not idx.is_cuda:
        raise ValueError('idx must be an int32 CUDA tensor')
    B, d = idx.numel(), mat_pi0.shape[1]
    out = torch.empty(B, d, dtype=torch.float32, device=mat_pi0.device)
    L.check(L.lib().mfg_gather_start(mat_pi0.data_ptr(), mat_pi0.shape[0], idx.data_ptr(), B, d, out.data_ptr(),
                                     _stream()), 'mfg_gather_start')
    return out


def draw_start(mat_pi0, B, seed, step, traj_offset=0, want_idx=False, want_pi0=True):
    """Start states of B trajectories drawn on the device (mfg_draw_start: Philox keyed by seed, the episode's first
    step, the global trajectory id).  Returns (idx int32 [B] | None, pi0 [B,d] | None)."""
    _chk_f32(mat_pi0, 'mat_pi0')
    d = mat_pi0.shape[1]
    idx = torch.empty(B, dtype=torch.int32, device=mat_pi0.device) if want_idx else None
    pi0 = torch.empty(B, d, dtype=torch.float32, device=mat_pi0.device) if want_pi0 else None
    L.check(L.lib().mfg_draw_start(mat_pi0.data_ptr(), mat_pi0.shape[0], int(B), d, int(seed), int(step), int(traj_offset),
                                   _ptr(idx), _ptr(pi0), _stream()), 'mfg_draw_start')
    return idx, pi0


def alpha(pi, theta, shift, want_alpha=True, want_deriv=True):
    _chk_f32(pi, 'pi'); _chk_f64(theta, 'theta')
    B, d = pi.shape
    a = torch.empty(B, d, d, dtype=torch.float64, device=pi.device) if want_alpha else None
    ad = torch.empty(B, d, d, dtype=torch.float64, device=pi.device) if want_deriv else None
    L.check(L.lib().mfg_alpha(pi.data_ptr(), B, d, theta.data_ptr(), float(shift), _ptr(a), _ptr(ad), _stream()),
            'mfg_alpha')
    return a, ad


def dirichlet_from_gamma(y):
    _chk_f32(y, 'y')
    B, d = y.shape[0], y.shape[-1]
    P = torch.empty_like(y)
    L.check(L.lib().mfg_dirichlet_from_gamma(y.data_ptr(), B, d, P.data_ptr(), _stream()), 'mfg_dirichlet_from_gamma')
    return P


def sample_dirichlet(pi, theta, shift, alpha_scale, seed, step=0, traj_offset=0, out=None, precision='mixed'):
    _chk_f32(pi, 'pi'); _chk_f64(theta, 'theta')
    B, d = pi.shape
    P = out if out is not None else torch.empty(B, d, d, dtype=torch.float32, device=pi.device)
    L.check(L.lib().mfg_sample_dirichlet(pi.data_ptr(), B, d, theta.data_ptr(), float(shift), float(alpha_scale),
                                         int(seed), int(step), int(traj_offset), L.PRECISIONS[precision], P.data_ptr(),
                                         _stream()),
            'mfg_sample_dirichlet')
    return P


def philox_raw(seed, first_ctr, c1, c2, c3, n, device):
    out = torch.empty(n, 4, dtype=torch.int32, device=device)
    L.check(L.lib().mfg_philox_raw(int(seed), int(first_ctr), int(c1), int(c2), int(c3), n, out.data_ptr(), _stream()),
            'mfg_philox_raw')
    return out


def step_given_P(pi, P, reward_kind=L.REWARD_MFG_AC2, want_reward=True):
    _chk_f32(pi, 'pi'); _chk_f32(P, 'P')
    B, d = pi.shape
    if P.shape != (B, d, d):
        raise ValueError('P must be [B,d,d]')
    pi_next = torch.empty_like(pi)
    reward = torch.empty(B, dtype=torch.float32, device=pi.device) if want_reward else None
    L.check(L.lib().mfg_step_given_P(pi.data_ptr(), P.data_ptr(), B, d, int(reward_kind), pi_next.data_ptr(),
                                     _ptr(reward), _stream()), 'mfg_step_given_P')
    return pi_next, reward


def value(pi, w):
    _chk_f32(pi, 'pi'); _chk_f64(w, 'w')
    B, d = pi.shape
    out = torch.empty(B, dtype=torch.float64, device=pi.device)
    L.check(L.lib().mfg_value(pi.data_ptr(), w.data_ptr(), B, d, out.data_ptr(), _stream()), 'mfg_value')
    return out


def features(pi):
    _chk_f32(pi, 'pi')
    B, d = pi.shape
    out = torch.empty(B, num_features(d), dtype=torch.float64, device=pi.device)
    L.check(L.lib().mfg_features(pi.data_ptr(), B, d, out.data_ptr(), _stream()), 'mfg_features')
    return out


def score(pi_alpha, P, theta, shift, precision='mixed'):
    _chk_f32(pi_alpha, 'pi'); _chk_f32(P, 'P'); _chk_f64(theta, 'theta')
    B, d = pi_alpha.shape
    g = torch.empty(B, dtype=torch.float64, device=P.device)
    L.check(L.lib().mfg_score(pi_alpha.data_ptr(), P.data_ptr(), B, d, theta.data_ptr(), float(shift),
                              L.PRECISIONS[precision], g.data_ptr(), _stream()), 'mfg_score')
    return g


def td_pg_accumulate(pi, pi_next, P, reward, w, theta, shift, gamma_or_discount, G=None, accumulate=False, ws=None,
                     precision='mixed', out=None):
    for t, n in ((pi, 'pi'), (pi_next, 'pi_next'), (P, 'P'), (reward, 'reward')):
        _chk_f32(t, n)
    _chk_f64(w, 'w'); _chk_f64(theta, 'theta')
    B, d = pi.shape
    F = num_features(d)
    if out is not None:
        delta, g = out
    else:
        delta = torch.empty(B, dtype=torch.float64, device=pi.device)
        g = torch.empty(B, dtype=torch.float64, device=pi.device)
    if G is None:
        G = torch.zeros(F + 3, dtype=torch.float64, device=pi.device)
    if ws is None:
        ws = workspace(B, d, pi.device)
    L.check(L.lib().mfg_td_pg_accumulate(pi.data_ptr(), pi_next.data_ptr(), P.data_ptr(), reward.data_ptr(),
                                         w.data_ptr(), theta.data_ptr(), float(shift), float(gamma_or_discount),
                                         B, d, L.PRECISIONS[precision], delta.data_ptr(), g.data_ptr(), G.data_ptr(),
                                         int(accumulate),
                                         ws.data_ptr(), ws.numel() * 8, _stream()), 'mfg_td_pg_accumulate')
    return delta, g, G


def apply_update(G, d, lr_critic, lr_actor, w, theta, reward_acc=None):
    """w, theta update from the batch sums; reward_acc (fp64 device scalar / 1-element view) += mean reward."""
    _chk_f64(G, 'G'); _chk_f64(w, 'w'); _chk_f64(theta, 'theta')
    L.check(L.lib().mfg_apply_update(G.data_ptr(), d, float(lr_critic), float(lr_actor), w.data_ptr(),
                                     theta.data_ptr(), _ptr(reward_acc), _stream()), 'mfg_apply_update')


def rollout(pi0, T, theta, shift, alpha_scale, w=None, gamma=1.0, reward_kind=L.REWARD_MFG_AC2, seed=0,
            first_step=0, traj_offset=0, td=True, write_P=False, discount_pow=False, G=None, accumulate=False,
            ws=None, out=None, precision='mixed'):
    """Fused T-step rollout.  Returns dict(pi_traj, pi_last, reward, delta, g, P, G)."""
    _chk_f32(pi0, 'pi0'); _chk_f64(theta, 'theta')
    B, d = pi0.shape
    dev = pi0.device
    o = out or {}
    pi_traj = o.get('pi_traj') if 'pi_traj' in o else torch.empty(B, T + 1, d, dtype=torch.float32, device=dev)
    pi_last = o.get('pi_last') if 'pi_last' in o else torch.empty(B, d, dtype=torch.float32, device=dev)
    ext = int(reward_kind) == L.REWARD_EXTERNAL
    if ext and G is not None:
        raise ValueError('external reward: the batch sums need the reward; call grad_accumulate(add_reward=True) afterwards')
    reward = None if ext else (o.get('reward') if 'reward' in o else torch.empty(B, T, dtype=torch.float32, device=dev))
    delta = g = None
    flags = 0
    if td:
        _chk_f64(w, 'w')
        flags |= L.ROLLOUT_TD
        delta = o.get('delta') if 'delta' in o else torch.empty(B, T, dtype=torch.float64, device=dev)
        g = o.get('g') if 'g' in o else torch.empty(B, T, dtype=torch.float64, device=dev)
        if not ext:
            if G is None:
                G = torch.zeros(num_features(d) + 3, dtype=torch.float64, device=dev)
            if ws is None:
                ws = workspace(B * T, d, dev)
    P = None
    if write_P:
        flags |= L.ROLLOUT_WRITE_P
        P = o.get('P') if 'P' in o else torch.empty(B, T, d, d, dtype=torch.float32, device=dev)
    if discount_pow:
        flags |= L.ROLLOUT_DISCOUNT_POW
    if L.PRECISIONS[precision] == L.PRECISION_F64:
        flags |= L.ROLLOUT_F64
    L.check(L.lib().mfg_rollout(pi0.data_ptr(), B, d, T, theta.data_ptr(), float(shift), float(alpha_scale),
                                _ptr(w) if td else None, float(gamma), int(reward_kind), int(seed), int(first_step),
                                int(traj_offset), flags, pi_traj.data_ptr(), _ptr(pi_last), _ptr(reward), _ptr(delta),
                                _ptr(g),
                                _ptr(P), _ptr(G) if (td and not ext) else None, int(accumulate),
                                _ptr(ws) if (td and not ext) else None,
                                ws.numel() * 8 if (td and not ext and ws is not None) else 0, _stream()), 'mfg_rollout')
    return {'pi_traj': pi_traj, 'pi_last': pi_last, 'reward': reward, 'delta': delta, 'g': g, 'P': P, 'G': G}


def train_rollout(mat_pi0, idx, T, theta, shift, alpha_scale, w, gamma, G, ws, bufs, lr_critic=0.0, lr_actor=0.0,
                  apply=False, reward_kind=L.REWARD_MFG_AC2, seed=0, first_step=0, traj_offset=0, discount_pow=False,
                  reward_acc=None, precision='mixed'):
    """One training update per episode: start-state gather (inside the kernel) + fused T-step TD rollout + batch sums
    [+ parameter update when apply=True (single GPU)].  bufs = dict(pi_traj [B,T+1,d], pi_last [B,d] | None,
    reward [B,T], delta [B,T], g [B,T]).  idx=None: the start rows are drawn inside the rollout kernel."""
    _chk_f32(mat_pi0, 'mat_pi0'); _chk_f64(theta, 'theta'); _chk_f64(w, 'w'); _chk_f64(G, 'G')
    if idx is not None and (idx.dtype != torch.int32 or not idx.is_cuda):
        raise ValueError('idx must be an int32 CUDA tensor (or None: start states drawn in the kernel)')
    B, d = bufs['pi_traj'].shape[0], mat_pi0.shape[1]
    if idx is not None and idx.numel() != B:
        raise ValueError('idx must have one entry per trajectory of the buffers')
    flags = (L.TRAIN_APPLY if apply else 0) | (L.ROLLOUT_DISCOUNT_POW if discount_pow else 0)
    if L.PRECISIONS[precision] == L.PRECISION_F64:
        flags |= L.ROLLOUT_F64
    L.check(L.lib().mfg_train_rollout(mat_pi0.data_ptr(), mat_pi0.shape[0], _ptr(idx), B, d, int(T), theta.data_ptr(),
                                      float(shift), float(alpha_scale), w.data_ptr(), float(gamma), int(reward_kind),
                                      int(seed), int(first_step), int(traj_offset), flags, float(lr_critic),
                                      float(lr_actor), bufs['pi_traj'].data_ptr(), _ptr(bufs.get('pi_last')),
                                      bufs['reward'].data_ptr(), bufs['delta'].data_ptr(), bufs['g'].data_ptr(),
                                      G.data_ptr(), _ptr(reward_acc), ws.data_ptr(), ws.numel() * 8, _stream()),
            'mfg_train_rollout')
    return bufs


def train_rollout_deferred(mat_pi0, idx, T, theta, w, pending, theta_out, w_out, shift, alpha_scale, gamma, G, ws, bufs,
                           reward_kind=L.REWARD_MFG_AC2, seed=0, first_step=0, traj_offset=0, discount_pow=False,
                           precision='mixed'):
    """Multi-rank update cycle without an update launch (mfg_train_rollout_deferred): `pending` = None or
    (G_all_reduced, lr_critic, lr_actor, reward_acc) of the PREVIOUS update, applied while this rollout stages its weights;
    the updated parameters land in (theta_out, w_out) -- other tensors than (theta, w).  G = this rank's sums afterwards."""
    _chk_f32(mat_pi0, 'mat_pi0'); _chk_f64(theta, 'theta'); _chk_f64(w, 'w'); _chk_f64(G, 'G')
    B, d = bufs['pi_traj'].shape[0], mat_pi0.shape[1]
    flags = L.ROLLOUT_DISCOUNT_POW if discount_pow else 0
    if L.PRECISIONS[precision] == L.PRECISION_F64:
        flags |= L.ROLLOUT_F64
    pG, plc, pla, pacc = pending if pending is not None else (None, 0.0, 0.0, None)
    L.check(L.lib().mfg_train_rollout_deferred(mat_pi0.data_ptr(), mat_pi0.shape[0], _ptr(idx), B, d, int(T), theta.data_ptr(),
                                               w.data_ptr(), _ptr(pG), float(plc), float(pla), _ptr(pacc), _ptr(theta_out),
                                               _ptr(w_out), float(shift), float(alpha_scale), float(gamma), int(reward_kind),
                                               int(seed), int(first_step), int(traj_offset), flags, bufs['pi_traj'].data_ptr(),
                                               _ptr(bufs.get('pi_last')), bufs['reward'].data_ptr(), bufs['delta'].data_ptr(),
                                               bufs['g'].data_ptr(), G.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream()),
            'mfg_train_rollout_deferred')
    return bufs


def train_rollouts(mat_pi0, T, episodes, first_episode, constant, theta, shift, alpha_scale, w, gamma, G, ws, bufs, lr_critic,
                   lr_actor, reward_kind=L.REWARD_MFG_AC2, seed=0, first_step=0, traj_offset=0, discount_pow=False,
                   reward_acc=None, precision='mixed'):
    """`episodes` training updates (one per episode) issued back to back by native code: start states drawn in the kernel,
    fused T-step rollout, batch sums, update with the reference's learning-rate schedule in episode numbers
    first_episode, first_episode + 1, ...  reward_acc: fp64 device array [episodes] (or an address), entry k += mean reward
    of episode k's update."""
    _chk_f32(mat_pi0, 'mat_pi0'); _chk_f64(theta, 'theta'); _chk_f64(w, 'w'); _chk_f64(G, 'G')
    B, d = bufs['pi_traj'].shape[0], mat_pi0.shape[1]
    flags = L.ROLLOUT_DISCOUNT_POW if discount_pow else 0
    if L.PRECISIONS[precision] == L.PRECISION_F64:
        flags |= L.ROLLOUT_F64
    L.check(L.lib().mfg_train_rollouts(mat_pi0.data_ptr(), mat_pi0.shape[0], B, d, int(T), int(episodes), int(first_episode),
                                       int(bool(constant)), theta.data_ptr(), float(shift), float(alpha_scale), w.data_ptr(),
                                       float(gamma), int(reward_kind), int(seed), int(first_step), int(traj_offset), flags,
                                       float(lr_critic), float(lr_actor), bufs['pi_traj'].data_ptr(), _ptr(bufs.get('pi_last')),
                                       bufs['reward'].data_ptr(), bufs['delta'].data_ptr(), bufs['g'].data_ptr(), G.data_ptr(),
                                       _ptr(reward_acc), ws.data_ptr(), ws.numel() * 8, _stream()), 'mfg_train_rollouts')
    return bufs


def train_rollouts_dist(comm, mat_pi0, T, episodes, first_episode, constant, theta, w, theta_alt, w_alt, shift, alpha_scale, gamma, G,
                        ws, bufs, lr_critic, lr_actor, reward_kind=L.REWARD_MFG_AC2, seed=0, first_step=0, traj_offset=0,
                        discount_pow=False, reward_acc=None, precision='mixed'):
    """`episodes` multi-GPU training updates issued natively (mfg_train_rollouts_dist): per episode the rollout kernel
    (previous update applied in its weight staging, start states drawn in the kernel), the batch sums and ONE RCCL all-reduce
    of G from the library itself; the last update is applied before returning, parameters end up in (theta, w).
    comm: address of a communicator from parallel.native_comm()."""
    _chk_f32(mat_pi0, 'mat_pi0'); _chk_f64(theta, 'theta'); _chk_f64(w, 'w'); _chk_f64(theta_alt, 'theta_alt'); _chk_f64(w_alt, 'w_alt')
    _chk_f64(G, 'G')
    B, d = bufs['pi_traj'].shape[0], mat_pi0.shape[1]
    flags = L.ROLLOUT_DISCOUNT_POW if discount_pow else 0
    if L.PRECISIONS[precision] == L.PRECISION_F64:
        flags |= L.ROLLOUT_F64
    L.check(L.lib().mfg_train_rollouts_dist(comm, mat_pi0.data_ptr(), mat_pi0.shape[0], B, d, int(T), int(episodes), int(first_episode),
                                            int(bool(constant)), theta.data_ptr(), w.data_ptr(), theta_alt.data_ptr(),
                                            w_alt.data_ptr(), float(shift), float(alpha_scale), float(gamma), int(reward_kind),
                                            int(seed), int(first_step), int(traj_offset), flags, float(lr_critic), float(lr_actor),
                                            bufs['pi_traj'].data_ptr(), _ptr(bufs.get('pi_last')), bufs['reward'].data_ptr(),
                                            bufs['delta'].data_ptr(), bufs['g'].data_ptr(), G.data_ptr(), _ptr(reward_acc),
                                            ws.data_ptr(), ws.numel() * 8, _stream()), 'mfg_train_rollouts_dist')
    return bufs


def grad_accumulate(pi, delta, g, reward, G, ws, T=1, stride_b=None, add_reward=False, accumulate=False):
    """Batch sums G (+)= [sum delta phi | sum delta g | sum r | N] over B*T samples; add_reward: delta += reward first
    (in place) -- the IRL step after the reward network has run."""
    d = pi.shape[-1]
    B = delta.numel() // T
    if stride_b is None:
        stride_b = (T + 1) * d if (pi.dim() == 3 and pi.shape[1] == T + 1) else T * d
    L.check(L.lib().mfg_grad_accumulate(pi.data_ptr(), int(stride_b), delta.data_ptr(), _ptr(g), _ptr(reward), B, int(T), d,
                                        int(bool(add_reward)), G.data_ptr(), int(bool(accumulate)), ws.data_ptr(),
                                        ws.numel() * 8, _stream()), 'mfg_grad_accumulate')
    return G


def grad_apply(pi, delta, g, reward, G, ws, lr_critic, lr_actor, w, theta, reward_acc=None, T=1, stride_b=None,
               add_reward=False):
    """grad_accumulate + apply_update in one call (single GPU): the update rides in the kernel that finishes the sums."""
    d = pi.shape[-1]
    B = delta.numel() // T
    if stride_b is None:
        stride_b = (T + 1) * d if (pi.dim() == 3 and pi.shape[1] == T + 1) else T * d
    L.check(L.lib().mfg_grad_apply(pi.data_ptr(), int(stride_b), delta.data_ptr(), _ptr(g), _ptr(reward), B, int(T), d,
                                   int(bool(add_reward)), G.data_ptr(), float(lr_critic), float(lr_actor), w.data_ptr(),
                                   theta.data_ptr(), _ptr(reward_acc), ws.data_ptr(), ws.numel() * 8, _stream()),
            'mfg_grad_apply')
    return G


def train_episode(pi, T, theta, shift, alpha_scale, w, gamma, lr_critic, lr_actor, G, ws, bufs, reward_kind=L.REWARD_MFG_AC2,
                  seed=0, first_step=0, traj_offset=0, reward_acc=None, precision='mixed'):
    """T env steps with the reference's per-step parameter updates, issued natively (single GPU).  `pi` [B,d] is
    updated in place to the final states; `bufs` = dict(scratch[B,d] f32, reward[B] f32, delta[B] f64, g[B] f64)."""
    _chk_f32(pi, 'pi'); _chk_f64(theta, 'theta'); _chk_f64(w, 'w')
    B, d = pi.shape
    L.check(L.lib().mfg_train_episode(pi.data_ptr(), bufs['scratch'].data_ptr(), B, d, int(T), theta.data_ptr(),
                                      float(shift), float(alpha_scale), w.data_ptr(), float(gamma), int(reward_kind),
                                      int(seed), int(first_step), int(traj_offset), L.PRECISIONS[precision],
                                      float(lr_critic), float(lr_actor), bufs['reward'].data_ptr(),
                                      bufs['delta'].data_ptr(), bufs['g'].data_ptr(), G.data_ptr(), _ptr(reward_acc),
                                      ws.data_ptr(), ws.numel() * 8, _stream()), 'mfg_train_episode')
    return pi


def train_episodes(mat_pi0, pi, T, episodes, first_episode, constant, theta, shift, alpha_scale, w, gamma, lr_critic, lr_actor,
                   G, ws, bufs, reward_kind=L.REWARD_MFG_AC2, seed=0, first_step=0, traj_offset=0, reward_acc=None,
                   precision='mixed'):
    """`episodes` x [start states drawn on the device into `pi` | T env steps with per-step updates], issued natively
    (single GPU); learning rates per episode as in train_rollouts.  `pi` [B,d] holds the last episode's final states."""
    _chk_f32(mat_pi0, 'mat_pi0'); _chk_f32(pi, 'pi'); _chk_f64(theta, 'theta'); _chk_f64(w, 'w')
    B, d = pi.shape
    L.check(L.lib().mfg_train_episodes(mat_pi0.data_ptr(), mat_pi0.shape[0], pi.data_ptr(), bufs['scratch'].data_ptr(), B, d,
                                       int(T), int(episodes), int(first_episode), int(bool(constant)), theta.data_ptr(),
                                       float(shift), float(alpha_scale), w.data_ptr(), float(gamma), int(reward_kind), int(seed),
                                       int(first_step), int(traj_offset), L.PRECISIONS[precision], float(lr_critic),
                                       float(lr_actor), bufs['reward'].data_ptr(), bufs['delta'].data_ptr(),
                                       bufs['g'].data_ptr(), G.data_ptr(), _ptr(reward_acc), ws.data_ptr(), ws.numel() * 8,
                                       _stream()), 'mfg_train_episodes')
    return pi


def reward_net_struct(net, dropout=None):
    """mfg_reward_net_t for a networks.RewardNet (device pointers of its parameters; keep the module alive while in use)."""
    if dropout is None:
        dropout = net.use_dropout and (net.dropout_always or net.training)
    st = L.RewardNetStruct()
    st.k1, st.f2, st.k2 = net.conv1.kernel_size[0], net.conv2.out_channels, net.conv2.kernel_size[0]
    st.n3, st.n4 = net.fc3.out_features, net.fc4.out_features
    for name, t in (('conv1_w', net.conv1.weight), ('conv1_b', net.conv1.bias), ('conv2_w', net.conv2.weight),
                    ('conv2_b', net.conv2.bias), ('fc3_w', net.fc3.weight), ('fc3_b', net.fc3.bias), ('fc4_w', net.fc4.weight),
                    ('fc4_b', net.fc4.bias), ('out_w', net.out.weight), ('out_b', net.out.bias)):
        if not t.is_contiguous():
            raise ValueError('reward net parameters must be contiguous')
        setattr(st, name, t.data_ptr())
    st.keep_prob = float(net.keep_prob) if dropout else 1.0
    return st


def train_episode_irl(pi, T, theta, shift, alpha_scale, w, gamma, lr_critic, lr_actor, net, G, ws, bufs, seed=0, first_step=0,
                      traj_offset=0, rn_seed=0, rn_call0=0, rn_sample_offset=0, reward_acc=None, precision='mixed', mat_pi0=None):
    """T env steps of AC_IRL.train with per-step updates, issued natively (single GPU): sample + transition + score,
    reward network, batch sums + update per step.  `pi` [B,d] is updated in place to the final states; `bufs` =
    dict(scratch [B,d] f32, P [B,d,d] f32, reward [B] f32, delta [B] f64, g [B] f64); `net` = networks.RewardNet.
    mat_pi0 [num_start,d]: the start states are drawn from this table inside the call (the draw of draw_start at step =
    first_step; `pi` is then an output only)."""
    import ctypes as C
    _chk_f32(pi, 'pi'); _chk_f64(theta, 'theta'); _chk_f64(w, 'w')
    B, d = pi.shape
    st = reward_net_struct(net)
    rest = (pi.data_ptr(), bufs['scratch'].data_ptr(), B, d, int(T), theta.data_ptr(), float(shift), float(alpha_scale),
            w.data_ptr(), float(gamma), int(seed), int(first_step), int(traj_offset), L.PRECISIONS[precision], float(lr_critic),
            float(lr_actor), C.byref(st), int(rn_seed) & 0xFFFFFFFFFFFFFFFF, int(rn_call0), int(rn_sample_offset),
            bufs['P'].data_ptr(), bufs['reward'].data_ptr(), bufs['delta'].data_ptr(), bufs['g'].data_ptr(), G.data_ptr(),
            _ptr(reward_acc), ws.data_ptr(), ws.numel() * 8, _stream())
    if mat_pi0 is None:
        L.check(L.lib().mfg_train_episode_irl(*rest), 'mfg_train_episode_irl')
    else:
        _chk_f32(mat_pi0, 'mat_pi0')
        if mat_pi0.dim() != 2 or mat_pi0.shape[1] != d:
            raise ValueError('train_episode_irl: mat_pi0 must be [num_start, %d]' % d)
        L.check(L.lib().mfg_train_episode_irl_draw(mat_pi0.data_ptr(), int(mat_pi0.shape[0]), *rest), 'mfg_train_episode_irl_draw')
    return pi


def train_rollout_irl(mat_pi0, idx, T, theta, shift, alpha_scale, w, gamma, lr_critic, lr_actor, net, G, ws, bufs, seed=0,
                      first_step=0, traj_offset=0, rn_key=0, rn_sample_offset=0, reward_acc=None, discount_pow=True, apply=True,
                      precision='mixed'):
    """One IRL update per episode issued natively (mfg_train_rollout_irl): fused rollout (start states drawn in the kernel when
    idx is None) | reward network over all B*T transitions | batch sums + update.  bufs: pi_traj [B,T+1,d], pi_last [B,d],
    P [B,T,d,d], reward [B*T] f32, delta / g [B*T] f64."""
    import ctypes as C
    _chk_f32(mat_pi0, 'mat_pi0'); _chk_f64(theta, 'theta'); _chk_f64(w, 'w'); _chk_f64(G, 'G')
    B, d = bufs['pi_traj'].shape[0], mat_pi0.shape[1]
    flags = (L.ROLLOUT_DISCOUNT_POW if discount_pow else 0) | (L.TRAIN_APPLY if apply else 0)
    if L.PRECISIONS[precision] == L.PRECISION_F64:
        flags |= L.ROLLOUT_F64
    st = reward_net_struct(net)
    L.check(L.lib().mfg_train_rollout_irl(mat_pi0.data_ptr(), mat_pi0.shape[0], _ptr(idx), B, d, int(T), theta.data_ptr(),
                                          float(shift), float(alpha_scale), w.data_ptr(), float(gamma), int(seed), int(first_step),
                                          int(traj_offset), flags, float(lr_critic), float(lr_actor), C.byref(st),
                                          int(rn_key) & 0xFFFFFFFFFFFFFFFF, int(rn_sample_offset), bufs['pi_traj'].data_ptr(),
                                          bufs['pi_last'].data_ptr(), bufs['P'].data_ptr(), bufs['reward'].data_ptr(),
                                          bufs['delta'].data_ptr(), bufs['g'].data_ptr(), G.data_ptr(), _ptr(reward_acc),
                                          ws.data_ptr(), ws.numel() * 8, _stream()), 'mfg_train_rollout_irl')
    return bufs


def episode_buffers(B, d, device):
    return {'scratch': torch.empty(B, d, dtype=torch.float32, device=device),
            'reward': torch.empty(B, dtype=torch.float32, device=device),
            'delta': torch.empty(B, dtype=torch.float64, device=device),
            'g': torch.empty(B, dtype=torch.float64, device=device)}


def jsd(p, q):
    _chk_f32(p, 'p'); _chk_f32(q, 'q')
    B, d = p.shape
    out = torch.empty(B, dtype=torch.float64, device=p.device)
    L.check(L.lib().mfg_jsd(p.data_ptr(), q.data_ptr(), B, d, out.data_ptr(), _stream()), 'mfg_jsd')
    return out


def policy_logpdf(pi, P, thetas, shift, alpha_scale=1.0, alpha_floor=0.0, p_floor=0.0):
    """log q_k(P_n | pi_n) [N,K] of the product-Dirichlet policy under K thetas (ac_irl.py:270-289, :324-379)."""
    _chk_f32(pi, 'pi'); _chk_f32(P, 'P'); _chk_f64(thetas, 'thetas')
    N, d = pi.shape
    if P.shape != (N, d, d):
        raise ValueError('P must be [N,d,d]')
    K = thetas.numel()
    out = torch.empty(N, K, dtype=torch.float64, device=pi.device)
    L.check(L.lib().mfg_policy_logpdf(pi.data_ptr(), P.data_ptr(), N, d, thetas.data_ptr(), K, float(shift),
                                      float(alpha_scale), float(alpha_floor), float(p_floor), out.data_ptr(), _stream()),
            'mfg_policy_logpdf')
    return out


def reward_net_supported(net) -> bool:
    """True when networks.RewardNet `net` fits the HIP forward kernel (d <= 32, f1 = 1, f2 <= 2, n_fc <= 32)."""
    k1, k2 = net.conv1.kernel_size[0], net.conv2.kernel_size[0]
    return (net.conv1.out_channels == 1 and net.conv2.out_channels <= 2 and net.d <= 32 and k1 % 2 == 1 and k2 % 2 == 1
            and k1 <= 7 and k2 <= 7 and net.fc3.out_features <= 32 and net.fc4.out_features <= 32
            and next(net.parameters()).dtype == torch.float32 and next(net.parameters()).is_cuda)


def reward_net_forward(net, state, action, dropout=None, seed=0, sample_offset=0):
    """r(state, action) [B] with the weights of a networks.RewardNet, one HIP launch (ac_irl.py:683 batched).
    dropout: None -> follow the module (active when the variant has dropout and dropout_always/training)."""
    _chk_f32(state, 'state'); _chk_f32(action, 'action')
    B, d = state.shape
    if dropout is None:
        dropout = net.use_dropout and (net.dropout_always or net.training)
    keep = float(net.keep_prob) if dropout else 1.0
    out = torch.empty(B, dtype=torch.float32, device=state.device)
    def P(t):
        if not t.is_contiguous():
            raise ValueError('reward net parameters must be contiguous')
        return t.data_ptr()
    L.check(L.lib().mfg_reward_net_forward(
        state.data_ptr(), action.data_ptr(), B, d, net.conv1.kernel_size[0], net.conv2.out_channels,
        net.conv2.kernel_size[0], net.fc3.out_features, net.fc4.out_features, P(net.conv1.weight), P(net.conv1.bias),
        P(net.conv2.weight), P(net.conv2.bias), P(net.fc3.weight), P(net.fc3.bias), P(net.fc4.weight), P(net.fc4.bias),
        P(net.out.weight), P(net.out.bias), keep, int(seed), int(sample_offset), out.data_ptr(), _stream()),
        'mfg_reward_net_forward')
    return out


def backward_value(P_seq, want_jsd=True):
    """mfg_synthetic backward recursion on actions P_seq [B,T,d,d]: returns V [B,T+1,d], diff_l1 [B,T], diff_jsd [B,T]|None."""
    _chk_f32(P_seq, 'P_seq')
    B, T, d, _ = P_seq.shape
    V = torch.empty(B, T + 1, d, dtype=torch.float64, device=P_seq.device)
    l1 = torch.empty(B, T, dtype=torch.float64, device=P_seq.device)
    js = torch.empty(B, T, dtype=torch.float64, device=P_seq.device) if want_jsd else None
    L.check(L.lib().mfg_backward_value(P_seq.data_ptr(), B, T, d, V.data_ptr(), l1.data_ptr(), _ptr(js), _stream()),
            'mfg_backward_value')
    return V, l1, js
