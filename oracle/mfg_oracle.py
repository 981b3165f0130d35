"""CPU oracle for the mean-field-game hot path.  TEST INFRASTRUCTURE ONLY.

This module is a float64 NumPy *restatement* of the reference's algorithm for the
hot path.  It is the checker for the HIP kernels, never the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package
``discrete_mean_field_game_amd`` must never import anything from ``oracle/``.

Pinning: ``oracle/gen_golden.py`` imports the unmodified reference
(``/root/reference/mfg_ac2.py``; ``ac_irl.py`` behind a stub ``tensorflow``) in the
build container and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors, including the reference's own
known-answer inputs (``test2.py:46-56``, ``:73-88``, ``:105-121``) and seeded
multi-episode ``train()`` traces.  The TF reward network (``networks.py``) cannot be
imported here (TensorFlow 1.x is absent) and the reference holds no numeric test
for it: that part is "parity unpinned" and lives in ``oracle/reward_net_oracle.py``.

Every function is batch-first: ``pi`` is ``(..., d)``, ``P`` is ``(..., d, d)``.  With no
leading batch dimension each function reduces to the reference's batch-1 math, in
the reference's operation order, so a seeded batch-1 run reproduces the reference's
trajectory bit for bit (checked in the golden tests).

Citations are ``file:line`` into the reference repository.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import numpy as np
from scipy import special
from scipy.stats import entropy

ZERO_GAMMA_REPLACEMENT = 1e-20   # mfg_ac2.py:244 / ac_irl.py:539
ZERO_P_REPLACEMENT = 1e-100      # mfg_ac2.py:369, :556-557
EPISODE_STEPS = 15               # mfg_ac2.py:478 / ac_irl.py:664


# --------------------------------------------------------------------------
# a5 / a9: integer bookkeeping
# --------------------------------------------------------------------------
def num_features(d: int) -> int:
    """F = d(d+1)/2 + d + 1.  mfg_ac2.py:175."""
    return (d + 1) * d // 2 + d + 1


def feature_index(i: int, j: int, d: int) -> int:
    """Position of the quadratic feature pi_i*pi_j (i <= j) inside phi(pi).

    The reference enumerates ``itertools.combinations_with_replacement(pi, 2)``
    (mfg_ac2.py:333), i.e. row-major upper triangle:
    k(i, j) = i*d - i(i-1)/2 + (j - i).
    """
    if i > j:
        i, j = j, i
    return i * d - (i * (i - 1)) // 2 + (j - i)


def feature_index_table(d: int) -> np.ndarray:
    """(d, d) int32 table of k(i, j) (symmetric), the index map the kernels use."""
    k = np.empty((d, d), dtype=np.int32)
    for i in range(d):
        for j in range(d):
            k[i, j] = feature_index(i, j, d)
    return k


def reorder(list_rows):
    """Stable sort of all rows by decreasing value of the first row.

    mfg_ac2.py:58-81 (``list.sort(reverse=True)`` is stable: ties keep index order).
    Returns (reordered rows, permutation).
    """
    row1 = list(list_rows[0])
    order = sorted(range(len(row1)), key=lambda i: row1[i], reverse=True)
    return [[row[j] for j in order] for row in list_rows], order


def parse_pi0_line(line: str, d: int):
    """First line of a trend_distribution_day file -> first d floats.  mfg_ac2.py:198."""
    return list(map(float, line.strip().split(' ')))[0:d]


# --------------------------------------------------------------------------
# a1: deterministic half of sample_action
# --------------------------------------------------------------------------
def calc_alpha(pi, theta, shift):
    """alpha_ij = ln(1 + exp(theta*(pi_j - pi_i - shift))).  mfg_ac2.py:225-228, ac_irl.py:521-524."""
    pi = np.asarray(pi, dtype=np.float64)
    temp = pi[..., None, :] - pi[..., :, None]          # temp_ij = pi_j - pi_i
    return np.log(1 + np.exp(theta * (temp - shift)))


def calc_alpha_deriv(pi, theta, shift):
    """d alpha_ij / d theta = x / (1 + exp(-theta*x)), x = pi_j - pi_i - shift.

    mfg_ac2.py:232-234, ac_irl.py:573-588.
    """
    pi = np.asarray(pi, dtype=np.float64)
    temp = pi[..., None, :] - pi[..., :, None]
    numerator = temp - shift
    denominator = 1 + np.exp((-theta) * numerator)
    return numerator / denominator


# --------------------------------------------------------------------------
# a2: stochastic half of sample_action
# --------------------------------------------------------------------------
def dirichlet_from_gamma(y):
    """Rows of gamma variates -> row-stochastic P.  mfg_ac2.py:244-249.

    Zeros are replaced by 1e-20 before normalising.  Input is not modified.
    """
    y = np.array(y, dtype=np.float64, copy=True)
    y[y == 0] = ZERO_GAMMA_REPLACEMENT
    return y / np.sum(y, axis=-1, keepdims=True)


def sample_action(pi, theta, shift, alpha_scale, rng=np.random, return_gamma=False):
    """Reference sampler for ONE trajectory: d sequential vector gamma draws.

    mfg_ac2.py:236-254 / ac_irl.py:526-548.  ``rng`` defaults to the process-global
    legacy ``np.random`` exactly like the reference, so a caller-side
    ``np.random.seed`` reproduces the reference's stream.
    """
    pi = np.asarray(pi, dtype=np.float64)
    d = pi.shape[-1]
    mat_alpha = calc_alpha(pi, theta, shift)
    y = np.zeros([d, d])
    for i in range(d):
        y[i] = rng.gamma(shape=mat_alpha[i, :] * alpha_scale, scale=1)
    P = dirichlet_from_gamma(y)
    if return_gamma:
        return P, y
    return P


# --------------------------------------------------------------------------
# a3 / a4: transition and reward
# --------------------------------------------------------------------------
def transition(P, pi):
    """pi'_j = sum_i P_ij pi_i.  mfg_ac2.py:497, ac_irl.py:679."""
    P = np.asarray(P, dtype=np.float64)
    pi = np.asarray(pi, dtype=np.float64)
    if P.ndim == 2:
        return np.transpose(P).dot(pi)
    return np.einsum('...ij,...i->...j', P, pi)


def calc_reward(P, pi):
    """R = pi . ((P*P) pi - ((P*P) 1) * pi).  mfg_ac2.py:274-279.

    Batch-1 input returns shape (1,) like the reference; batched returns (B,).
    """
    P = np.asarray(P, dtype=np.float64)
    pi = np.asarray(pi, dtype=np.float64)
    d = pi.shape[-1]
    if P.ndim == 2:
        P_squared = P * P
        v1 = P_squared.dot(pi.reshape(d, 1))
        v2 = P_squared.dot(np.ones([d, 1])) * pi.reshape(d, 1)
        return pi.dot(v1 - v2)
    P_squared = P * P
    v1 = np.einsum('...ij,...j->...i', P_squared, pi)
    v2 = np.sum(P_squared, axis=-1) * pi
    return np.sum(pi * (v1 - v2), axis=-1)


def calc_reward_synthetic(P, pi):
    """mfg_synthetic variant R = -1/2 sum_i pi_i ||P_i||^2.  mfg_synthetic.py:249-265."""
    P = np.asarray(P, dtype=np.float64)
    pi = np.asarray(pi, dtype=np.float64)
    return -0.5 * np.sum(pi * np.sum(P * P, axis=-1), axis=-1)


# --------------------------------------------------------------------------
# a5: features and value
# --------------------------------------------------------------------------
def calc_features(pi):
    """phi(pi) = [pi_i pi_j (i<=j, row-major upper triangle), pi_1..pi_d, 1].

    mfg_ac2.py:325-344 (the docstring at :171 states another order; the code wins).
    """
    pi = np.asarray(pi, dtype=np.float64)
    d = pi.shape[-1]
    iu, ju = np.triu_indices(d)
    quad = pi[..., iu] * pi[..., ju]
    ones = np.ones(pi.shape[:-1] + (1,))
    return np.concatenate([quad, pi, ones], axis=-1)


def calc_value(pi, w):
    """V(pi; w) = phi(pi) . w.  mfg_ac2.py:290-322."""
    return calc_features(pi).dot(np.asarray(w, dtype=np.float64))


# --------------------------------------------------------------------------
# a7: score of the product-Dirichlet policy w.r.t. theta
# --------------------------------------------------------------------------
def calc_gradient(P, pi, theta, shift, mat_alpha=None, mat_alpha_deriv=None):
    """g = sum_ij (-psi(alpha_ij) + psi(sum_j alpha_ij) + ln P_ij) * alpha'_ij.

    mfg_ac2.py:347-381 / ac_irl.py:591-624.  alpha is the UNSCALED concentration
    (no alpha_scale).  The reference reads ``self.mat_alpha`` left behind by the last
    ``sample_action`` call; pass ``mat_alpha`` to reproduce a stale-alpha call,
    otherwise it is recomputed from ``pi``.  Zeros of P count as 1e-100 (:369); the
    input array is not modified here (the reference mutates it in place).
    """
    P = np.array(P, dtype=np.float64, copy=True)
    if mat_alpha is None:
        mat_alpha = calc_alpha(pi, theta, shift)
    if mat_alpha_deriv is None:
        mat_alpha_deriv = calc_alpha_deriv(pi, theta, shift)
    mat1 = special.digamma(mat_alpha)
    mat2 = special.digamma(np.ones(mat_alpha.shape) * np.sum(mat_alpha, axis=-1, keepdims=True))
    P[P == 0] = ZERO_P_REPLACEMENT
    mat3 = np.log(P)
    return np.sum((-mat1 + mat2 + mat3) * mat_alpha_deriv, axis=(-2, -1))


# --------------------------------------------------------------------------
# f3: mfg_synthetic backward value recursion and its consistency metrics
# --------------------------------------------------------------------------
def calc_reward_vector(P):
    """v_i = -1/2 ||P_i||^2.  mfg_synthetic.py:726-738."""
    P = np.asarray(P, dtype=np.float64)
    return -0.5 * np.sum(P * P, axis=-1)


def backward_value(actions):
    """V^n = r^n + P^n V^{n+1}, V^T = 0, for actions[..., T, d, d] -> V[..., T+1, d].
    mfg_synthetic.py:768-774 (its mat_V is [d, 16]; here time-major)."""
    actions = np.asarray(actions, dtype=np.float64)
    T, d = actions.shape[-3], actions.shape[-1]
    V = np.zeros(actions.shape[:-3] + (T + 1, d))
    for n in range(T - 1, -1, -1):
        V[..., n, :] = calc_reward_vector(actions[..., n, :, :]) + np.einsum('...ij,...j->...i', actions[..., n, :, :],
                                                                              V[..., n + 1, :])
    return V


def value_implied_rows(V_n):
    """Row i of the matrix implied by the value function: V_j - V_i off the diagonal,
    1 - (sum V - d V_i) on it.  mfg_synthetic.py:784-790."""
    V_n = np.asarray(V_n, dtype=np.float64)
    d = V_n.shape[-1]
    M = V_n[..., None, :] - V_n[..., :, None]
    diag = 1 - (np.sum(V_n, axis=-1, keepdims=True) - d * V_n)
    idx = np.arange(d)
    M[..., idx, idx] = diag
    return M


def JSD_synthetic(P, Q):
    """mfg_synthetic.py:528-547: like JSD but every entry <= 0 becomes 1e-100."""
    P = np.array(P, dtype=np.float64, copy=True)
    Q = np.array(Q, dtype=np.float64, copy=True)
    P[P <= 0] = ZERO_P_REPLACEMENT
    Q[Q <= 0] = ZERO_P_REPLACEMENT
    M = 0.5 * (P + Q)
    return 0.5 * (entropy(P, M, axis=-1) + entropy(Q, M, axis=-1))


def evaluate_synthetic_diffs(actions):
    """Per (trajectory, hour) metrics of evaluate_synthetic (sum_ij |P_ij - value_ij|) and of
    evaluate_synthetic_JSD (sum_i JSD(P_i, implied row i)).  Returns (V, l1[..., T], jsd[..., T])."""
    actions = np.asarray(actions, dtype=np.float64)
    V = backward_value(actions)
    M = value_implied_rows(V[..., :-1, :])
    l1 = np.sum(np.abs(actions - M), axis=(-2, -1))
    jsd = np.sum(JSD_synthetic(actions, M), axis=-1)
    return V, l1, jsd


# --------------------------------------------------------------------------
# a11: Jensen-Shannon divergence
# --------------------------------------------------------------------------
def JSD(P, Q):
    """0.5*(KL(P||M) + KL(Q||M)), M = (P+Q)/2, zeros -> 1e-100.  mfg_ac2.py:546-563.

    ``scipy.stats.entropy`` renormalises each argument.  Inputs are not modified
    (the reference replaces zeros in place).
    """
    P = np.array(P, dtype=np.float64, copy=True)
    Q = np.array(Q, dtype=np.float64, copy=True)
    P[P == 0] = ZERO_P_REPLACEMENT
    Q[Q == 0] = ZERO_P_REPLACEMENT
    M = 0.5 * (P + Q)
    return 0.5 * (entropy(P, M, axis=-1) + entropy(Q, M, axis=-1))


# --------------------------------------------------------------------------
# a6 / a8: learning-rate schedules and one TD / actor-critic update
# --------------------------------------------------------------------------
def lr_scales(episode: int, constant, variant: str = 'mfg_ac2'):
    """Multipliers applied to (lr_critic, lr_actor) in ``episode``.

    mfg_ac2.py:511-522 (episodes 0-indexed) and ac_irl.py:697-708 (1-indexed, same
    formula written in terms of its own ``episode``).
    """
    if constant:
        return 1.0, 1.0
    return 1.0 / (episode + 1), 1.0 / ((episode + 1) * np.log(np.log(episode + 20)))


def ac_step(pi, P, reward, w, theta, shift, gamma_or_discount, lr_c, lr_a,
            mat_alpha=None, mat_alpha_deriv=None):
    """One reference actor-critic update given (pi, P, reward).

    mfg_ac2.py:497-522 / ac_irl.py:679-708.  ``w`` is (F,1).  Returns
    (pi_next, delta, gradient, w_new, theta_new).
    """
    pi_next = transition(P, pi)
    vec_features_next = calc_features(pi_next)
    vec_features = calc_features(pi)
    delta = reward + gamma_or_discount * (vec_features_next.dot(w)) - (vec_features.dot(w))
    length = len(vec_features)
    w_new = w + lr_c * delta * vec_features.reshape(length, 1)
    gradient = calc_gradient(P, pi, theta, shift, mat_alpha, mat_alpha_deriv)
    theta_new = theta + lr_a * delta * gradient
    return pi_next, delta, gradient, w_new, theta_new


# --------------------------------------------------------------------------
# a9: train() control flow, batch 1, reference RNG order
# --------------------------------------------------------------------------
def train_mfg_ac2(mat_pi0, w, theta, shift, alpha_scale, num_episodes, gamma=1,
                  constant=0, lr_critic=0.1, lr_actor=0.001, rng=np.random,
                  reward_fn: Optional[Callable] = None, record=True):
    """Restatement of ``mfg_ac2.actor_critic.train`` (mfg_ac2.py:448-539), batch 1.

    Consumes ``rng`` in the reference's order: one ``randint`` per episode (:466)
    then d ``gamma`` vector draws per step (:242).  Returns a dict with the final
    (theta, w) and per-step traces used for parity fixtures.
    """
    mat_pi0 = np.asarray(mat_pi0, dtype=np.float64)
    d = mat_pi0.shape[1]
    num_start = mat_pi0.shape[0]
    w = np.array(w, dtype=np.float64, copy=True).reshape(-1, 1)
    tr = {'idx_row': [], 'theta': [], 'delta': [], 'reward': [], 'gradient': [],
          'total_reward': [], 'pi_final': [], 'P': [], 'pi': []}
    for episode in range(num_episodes):
        idx_row = rng.randint(num_start)
        pi = mat_pi0[idx_row, :]
        total_reward = 0
        for _ in range(EPISODE_STEPS):
            P = sample_action(pi, theta, shift, alpha_scale, rng)
            if reward_fn is None:
                reward = calc_reward(P, pi)
            else:
                reward = reward_fn(pi, P)
            sc, sa = lr_scales(episode, constant == 1)
            if record:
                tr['pi'].append(np.array(pi))
                tr['P'].append(P.copy())
            pi_next, delta, gradient, w, theta = ac_step(
                pi, P, reward, w, theta, shift, gamma, lr_critic * sc, lr_actor * sa)
            pi = pi_next
            total_reward += reward
            if record:
                tr['theta'].append(np.ravel(theta)[0])
                tr['delta'].append(np.ravel(delta)[0])
                tr['reward'].append(np.ravel(reward)[0])
                tr['gradient'].append(float(gradient))
        tr['idx_row'].append(idx_row)
        tr['total_reward'].append(np.ravel(total_reward)[0])
        tr['pi_final'].append(np.array(pi))
    out = {k: np.array(v) for k, v in tr.items()}
    out['idx_row'] = out['idx_row'].astype(np.int64)
    out['theta_final'] = np.ravel(theta)[0]
    out['w_final'] = w
    return out


def train_log_lines(out, consecutive):
    """Text the reference appends to file_theta / file_pi / file_reward with write_file=1 (mfg_ac2.py:441-445,
    :530-539) for a finished ``train_mfg_ac2`` trace: every ``consecutive``-th episode (from 0) one line each with theta
    ('%.5e'), the episode's final pi ('%.3e') and sum(returns since the last line) / consecutive ('%.3e')."""
    import io

    def line(vec, fmt):
        return ','.join(fmt % v for v in np.ravel(vec)) + '\n'
    th = pi = rw = ''
    acc = []
    for ep in range(len(out['total_reward'])):
        acc.append(out['total_reward'][ep])
        if ep % consecutive == 0:
            th += line([out['theta'][EPISODE_STEPS * (ep + 1) - 1]], '%.5e')
            pi += line(out['pi_final'][ep], '%.3e')
            rw += line([sum(acc) / consecutive], '%.3e')
            acc = []
    return th, pi, rw


def train_ac_irl(mat_pi0, w, theta, shift, alpha_scale, max_episodes, reward_fn,
                 stop_criteria=0.01, gamma=1, constant=False, lr_critic=0.1,
                 lr_actor=0.001, rng=np.random):
    """Restatement of ``ac_irl.AC_IRL.train`` (ac_irl.py:634-732), batch 1.

    Differences from mfg_ac2 kept on purpose: episodes are 1-indexed (:649), the
    bootstrap term uses the running ``discount = gamma**t`` (:691, :710), reward
    comes from ``reward_fn(pi, P)`` (the TF net at :683), early stop on
    |theta - prev_theta| < stop_criteria (:726).  Returns traces + episodes run.
    """
    mat_pi0 = np.asarray(mat_pi0, dtype=np.float64)
    num_start = mat_pi0.shape[0]
    w = np.array(w, dtype=np.float64, copy=True).reshape(-1, 1)
    tr = {'idx_row': [], 'theta': [], 'delta': [], 'reward': [], 'gradient': [], 'total_reward': []}
    prev_theta = theta
    episode = 0
    for episode in range(1, max_episodes + 1):
        idx_row = rng.randint(num_start)
        pi = mat_pi0[idx_row, :]
        discount = 1
        total_reward = 0
        for _ in range(EPISODE_STEPS):
            P = sample_action(pi, theta, shift, alpha_scale, rng)
            reward = reward_fn(pi, P)
            sc, sa = lr_scales(episode, constant)
            pi_next, delta, gradient, w, theta = ac_step(
                pi, P, reward, w, theta, shift, discount, lr_critic * sc, lr_actor * sa)
            discount = discount * gamma
            pi = pi_next
            total_reward += reward
            tr['theta'].append(np.ravel(theta)[0])
            tr['delta'].append(np.ravel(delta)[0])
            tr['reward'].append(np.ravel(reward)[0])
            tr['gradient'].append(float(gradient))
        tr['idx_row'].append(idx_row)
        tr['total_reward'].append(np.ravel(total_reward)[0])
        if stop_criteria != -1 and abs(theta - prev_theta) < stop_criteria:
            break
        prev_theta = theta
    out = {k: np.array(v) for k, v in tr.items()}
    out['idx_row'] = out['idx_row'].astype(np.int64)
    out['theta_final'] = np.ravel(theta)[0]
    out['w_final'] = w
    out['episodes_run'] = episode
    return out


# --------------------------------------------------------------------------
# a10: trajectory generation
# --------------------------------------------------------------------------
def generate_trajectory(pi0, total_hours, theta, shift, alpha_scale, rng=np.random):
    """mfg_ac2.py:566-592: rows pi^0..pi^{total_hours-1}."""
    pi = np.asarray(pi0, dtype=np.float64)
    d = pi.shape[-1]
    mat = np.zeros([total_hours, d])
    mat[0] = pi
    for hour in range(1, total_hours):
        P = sample_action(pi, theta, shift, alpha_scale, rng)
        pi = transition(P, pi)
        mat[hour] = pi
    return mat


def generate_trajectories(n, mat_pi0, theta, shift, alpha_scale, rng=np.random):
    """ac_irl.py:735-767: list of n trajectories, each 15 (pi, P) pairs."""
    mat_pi0 = np.asarray(mat_pi0, dtype=np.float64)
    out = []
    for _ in range(n):
        idx_row = rng.randint(mat_pi0.shape[0])
        pi = mat_pi0[idx_row, :]
        traj = []
        for _hour in range(1, 16):
            P = sample_action(pi, theta, shift, alpha_scale, rng)
            traj.append((pi, P))
            pi = transition(P, pi)
        out.append(traj)
    return out


# --------------------------------------------------------------------------
# f2: evaluate / gridsearch (mfg_ac2.py:595-689)
# --------------------------------------------------------------------------
def evaluate(list_empirical, theta, shift, alpha_scale, d, episode_length=16, rng=np.random):
    """Restatement of ``actor_critic.evaluate`` (mfg_ac2.py:595-670) on the test matrices in the order the reference
    met them (its ``os.listdir`` order: every test trajectory is generated to completion before the next file is read,
    which fixes the np.random consumption order).  Returns the eight statistics
    (mean_l1_final, std_l1_final, mean_l1_mean, std_l1_mean, mean_JSD_final, std_JSD_final, mean_JSD_mean, std_JSD_mean).
    """
    n = len(list_empirical)
    a_l1f, a_l1m, a_jf, a_jm = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    for idx, mat in enumerate(list_empirical):
        mat_empirical = np.array(mat, dtype=np.float64)[:, 0:d]
        traj = generate_trajectory(mat_empirical[0], episode_length, theta, shift, alpha_scale, rng)
        a_l1f[idx] = np.sum(np.abs(traj[-1] - mat_empirical[-1]))                   # norm(., ord=1), :636
        diff = mat_empirical - traj
        a_l1m[idx] = np.mean(np.sum(np.abs(diff), axis=1))                          # :640-641
        a_jf[idx] = JSD(traj[-1], mat_empirical[-1])                                # :645
        jm = 0
        for idx2 in range(episode_length):                                          # :649-653
            jm += JSD(mat_empirical[idx2], traj[idx2])
        a_jm[idx] = jm / episode_length
    return (np.mean(a_l1f), np.std(a_l1f), np.mean(a_l1m), np.std(a_l1m),
            np.mean(a_jf), np.std(a_jf), np.mean(a_jm), np.std(a_jm))


def format_eval_line(theta, shift, alpha_scale, stats):
    """The CSV line evaluate() appends (mfg_ac2.py:668)."""
    return "%f,%f,%f,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e,%.3e\n" % ((theta, shift, alpha_scale) + tuple(stats))


def gridsearch(list_empirical, theta_range, shift_range, alpha_range, d, rng=np.random):
    """mfg_ac2.py:673-689: evaluate every (theta, shift, alpha_scale) in nested-loop order, keep the argmin (ties go
    to the LATER point: ``<=``) of mean_l1_final, mean_l1_mean, mean_JSD_final, mean_JSD_mean.  Returns
    (list_tuples, csv_text)."""
    list_tuples = [[100, 0, 0, 0], [100, 0, 0, 0], [100, 0, 0, 0], [100, 0, 0, 0]]
    text = ''
    for theta in theta_range:
        for shift in shift_range:
            for alpha_scale in alpha_range:
                st = evaluate(list_empirical, theta, shift, alpha_scale, d, 16, rng)
                text += format_eval_line(theta, shift, alpha_scale, st)
                result = (st[0], st[2], st[4], st[6])
                for idx in range(4):
                    if result[idx] <= list_tuples[idx][0]:
                        list_tuples[idx] = [result[idx], theta, shift, alpha_scale]
    return list_tuples, text


def normalize_rows_text(matrix):
    """mfg_ac2.py:116-137: rows divided by their sums, written as '%.3e' space separated text."""
    import io
    matrix = np.asarray(matrix, dtype=np.float64)
    matrix = matrix / np.sum(matrix, axis=1, keepdims=True)
    buf = io.BytesIO()
    np.savetxt(buf, matrix, fmt='%.3e', delimiter=' ')
    return buf.getvalue().decode()


def read_demonstration_day(state_text, action_text, d, dim_action=20):
    """ac_irl.py:164-200 for one day: 15 (state [d], action [d,d]) pairs from the text of a state file (16 rows) and
    of an action file (15 blocks of dim_action rows, blank lines skipped by the parser)."""
    states = np.array([[float(v) for v in line.split(' ')] for line in state_text.strip().split('\n')])
    rows = [line for line in action_text.split('\n') if line.strip() != '']
    actions = np.array([[float(v) for v in line.split(' ')] for line in rows])
    return [(states[hour, 0:d], actions[hour * dim_action:(hour * dim_action + d), 0:d]) for hour in range(15)]


# --------------------------------------------------------------------------
# Batched semantics of the new framework (B trajectories in lock-step)
# --------------------------------------------------------------------------
def batched_td_pg(pi, pi_next, P, reward, w, theta, shift, gamma_or_discount):
    """Batched a5-a8 pieces on given inputs: returns delta[B], g[B], G_w[F], G_theta, sum_reward.

    G_w = sum_b delta_b phi(pi_b), G_theta = sum_b delta_b g_b (plain sums; the caller
    divides by the global batch).  With B = 1 this is exactly the increment the
    reference applies (mfg_ac2.py:505-522) divided by the learning rate.
    """
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    phi = calc_features(pi)
    phi_next = calc_features(pi_next)
    delta = np.asarray(reward, dtype=np.float64) + gamma_or_discount * phi_next.dot(w) - phi.dot(w)
    g = calc_gradient(P, pi, theta, shift)
    G_w = np.einsum('b,bf->f', delta, phi)
    G_theta = float(np.sum(delta * g))
    return delta, g, G_w, G_theta, float(np.sum(np.asarray(reward, dtype=np.float64)))


def batched_rollout_given_P(pi0, P_seq, w, theta, shift, gamma=1.0, variant='mfg_ac2',
                            storage_dtype=np.float32, reward_seq=None):
    """T-step lock-step rollout with FIXED (theta, w) on given actions P_seq[B,T,d,d].

    Mirrors the fused kernel's 'rollout' mode: pi is rounded to the storage dtype at
    every step (the kernel keeps pi in fp32 registers), gradients are summed over all
    B*T transitions.  Returns pi_traj[B,T+1,d], reward[B,T], delta[B,T], g[B,T],
    G_w[F], G_theta.
    """
    pi = np.asarray(pi0, dtype=storage_dtype).astype(np.float64)
    B, T = P_seq.shape[0], P_seq.shape[1]
    d = pi.shape[-1]
    F = num_features(d)
    traj = np.zeros((B, T + 1, d))
    traj[:, 0] = pi
    R = np.zeros((B, T)); D = np.zeros((B, T)); G = np.zeros((B, T))
    G_w = np.zeros(F); G_theta = 0.0
    discount = 1.0
    for t in range(T):
        P = np.asarray(P_seq[:, t], dtype=np.float64)
        pi_next = transition(P, pi).astype(storage_dtype).astype(np.float64)
        r = calc_reward(P, pi) if reward_seq is None else np.asarray(reward_seq[:, t], np.float64)
        gd = gamma if variant == 'mfg_ac2' else discount
        delta, g, gw, gt, _ = batched_td_pg(pi, pi_next, P, r, w, theta, shift, gd)
        R[:, t] = r; D[:, t] = delta; G[:, t] = g
        G_w += gw; G_theta += gt
        discount *= gamma
        pi = pi_next
        traj[:, t + 1] = pi
    return traj, R, D, G, G_w, G_theta


def cpu_baseline_steps(d, n_steps, theta=8.86349, shift=0.16, alpha_scale=12000, seed=0):
    """Reference-faithful CPU leg for bench.py: batch 1, sequential, full train() step.

    Runs ``n_steps // 15`` episodes of ``train_mfg_ac2`` on synthetic Dirichlet(1)
    start states and returns (env_steps_done, seconds).
    """
    import time
    rs = np.random.RandomState(seed)
    mat_pi0 = rs.dirichlet(np.ones(d), size=64)
    w = rs.rand(num_features(d), 1)
    episodes = max(1, n_steps // EPISODE_STEPS)
    t0 = time.perf_counter()
    train_mfg_ac2(mat_pi0, w, theta, shift, alpha_scale, episodes, rng=rs, record=False)
    return episodes * EPISODE_STEPS, time.perf_counter() - t0


# --------------------------------------------------------------------------
# f1 (optional importance weights of the max-ent loss; TensorFlow code in the reference, disabled at
# ac_irl.py:404-405 -- "parity unpinned": checked against scipy.stats.dirichlet and hand-derived values only)

def policy_logpdf(pi, P, thetas, shift, alpha_scale=1.0, alpha_floor=0.0, p_floor=0.0):
    """log q_k(P_n | pi_n) for N pairs under K policies -> [N, K].

    calc_pdf_action (ac_irl.py:270-289): mat_alpha = log(1 + exp(theta (pi_j - pi_i - shift))), pdf = prod_rows
    Dirichlet(mat_alpha[row]).prob(action[row]).  calc_z (:324-379) lower-bounds alpha by 1+1e-6 (:359) and
    multiplies the per-row densities over topics and time after dividing by a normaliser; here the logs are summed.
    """
    pi = np.asarray(pi, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    thetas = np.asarray(thetas, dtype=np.float64).reshape(-1)
    out = np.zeros((pi.shape[0], thetas.size))
    Pc = np.maximum(P, p_floor)
    with np.errstate(divide='ignore'):
        lnP = np.log(Pc)
    for k, th in enumerate(thetas):
        x = pi[:, None, :] - pi[:, :, None] - shift                   # x[n,i,j] = pi_j - pi_i - shift
        al = np.maximum(alpha_scale * np.log1p(np.exp(th * x)), alpha_floor)
        out[:, k] = np.sum(special.gammaln(al.sum(-1)) - special.gammaln(al).sum(-1) + ((al - 1.0) * lnP).sum(-1), axis=-1)
    return out


def calc_z(pi_traj, P_traj, thetas, shift, num_start_samples, alpha_floor=1.0 + 1e-6):
    """z_j = [1/K sum_k q_k(traj_j)]^-1 with q_k(traj) = Pr(s_1) prod_t q_k(a_t; s_t) (ac_irl.py:292-321, :374-379), in
    log space: returns log z_j for pi_traj [M,T,d], P_traj [M,T,d,d]."""
    M, T, d = pi_traj.shape
    lq = policy_logpdf(pi_traj.reshape(M * T, d), P_traj.reshape(M * T, d, d), thetas, shift, 1.0, alpha_floor)
    lq = lq.reshape(M, T, -1).sum(1) - np.log(num_start_samples)      # [M,K]
    K = lq.shape[1]
    return np.log(K) - special.logsumexp(lq, axis=1)
