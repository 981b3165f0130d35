#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference in this container.

TEST INFRASTRUCTURE.  Usage (build container only; /root/reference does not exist on
the GPU box and nothing at test time reads it):

    python oracle/gen_golden.py            # writes tests/golden/*.npz

How the reference is driven:
  * each reference module is imported in its own subprocess, because importing
    mfg_ac2 installs ``warnings.filterwarnings('error')`` process-wide
    (mfg_ac2.py:21) and a second module then fails on invalid-escape docstrings;
  * the constructor reads ``cwd/train_normalized_round2/trend_distribution_day%d.csv``
    (mfg_ac2.py:39), so we chdir to a scratch dir holding synthetic files written in
    the reference's own format (``%.3e`` space separated, mfg_ac2.py:137);
  * ac_irl imports TensorFlow 1.x (absent): an empty stub module is registered as
    ``tensorflow``, the object is built with ``object.__new__`` and given a fake
    ``sess`` whose ``run`` evaluates a closed-form reward, so ``AC_IRL.train`` /
    ``generate_trajectories`` execute the reference's own NumPy code and control flow.

Only inputs/outputs (data) are stored; no reference source text is copied.
"""
import argparse
import io
import contextlib
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def write_start_states(root, d, n_days, seed, subdir='train_normalized_round2', first_day=1):
    """Synthetic day files: 16 rows x d, Dirichlet(1) rows, '%.3e' text."""
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, subdir), exist_ok=True)
    for day in range(first_day, first_day + n_days):
        m = rs.dirichlet(np.ones(d), size=16)
        np.savetxt(os.path.join(root, subdir, 'trend_distribution_day%d.csv' % day), m,
                   fmt='%.3e', delimiter=' ')


def test_pis(d, rs):
    """State vectors used for per-function fixtures."""
    out = []
    for conc in (0.1, 1.0, 10.0):
        out.append(rs.dirichlet(np.ones(d) * conc))
    if d >= 2:
        p = np.ones(d) * 0.1 / (d - 1)      # degenerate start of test2.py:208-210
        p[0] = 0.9
        out.append(p)
    p = np.zeros(d)                        # one-hot, test2.py:63-64
    p[0] = 1.0
    out.append(p)
    return np.array(out)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def read_text(path):
    with open(path) as f:
        return f.read()


# --------------------------------------------------------------------------
def part_mfg_ac2(scratch):
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    os.chdir(scratch)
    write_start_states(scratch, 21, 4, seed=0)
    import mfg_ac2  # noqa: the reference, unmodified

    # ---- known-answer inputs held by the reference's own test scripts -------------
    kat = {}
    ac = mfg_ac2.actor_critic()
    P = np.array([[1, 3, 3], [4, 5, 6], [7, 8, 9]])
    pi = np.array([0.1, 0.2, 0.7])
    kat['reward_P'] = P.astype(np.float64)
    kat['reward_pi'] = pi
    kat['reward_out'] = ac.calc_reward(P, pi, 3)                      # test2.py:46-56
    ac3 = mfg_ac2.actor_critic(d=3)
    ac3.w = np.ones(10)
    kat['value_pi'] = pi
    kat['value_out'] = np.array(ac3.calc_value(pi))                   # test2.py:73-88
    kat['value_features'] = ac3.calc_features(pi)
    kat['jsd_p'] = np.array([.5, .5, 0.])
    kat['jsd_q'] = np.array([.1, .2, .7])
    kat['jsd_out'] = np.array(ac.JSD(np.array([.5, .5, 0.]), np.array([.1, .2, .7])))
    ac4 = mfg_ac2.actor_critic(theta=10, shift=0.4, d=4)              # test2.py:6,105-121
    pi4 = np.array([0.7, 0.09, 0.01, 0.2])
    np.random.seed(42)
    P4 = ac4.sample_action(pi4)
    kat['grad_pi'] = pi4
    kat['grad_P'] = P4.copy()
    kat['grad_alpha'] = ac4.mat_alpha.copy()
    kat['grad_alpha_deriv'] = ac4.mat_alpha_deriv.copy()
    kat['grad_basic'] = np.array(ac4.calc_gradient_basic(P4, pi4))
    kat['grad_loop'] = np.array(ac4.calc_gradient(P4, pi4))
    kat['grad_vectorized'] = np.array(ac4.calc_gradient_vectorized(P4, pi4))
    kat['grad_reward'] = ac4.calc_reward(P4, pi4, 4)
    kat['grad_pi_next'] = np.transpose(P4).dot(pi4)
    # test_action invariants (test2.py:14-32) are properties, checked in the tests.
    np.savez_compressed(os.path.join(OUT, 'kat_mfg_ac2.npz'), **kat)

    # ---- per-function vectors at several (d, theta, shift, alpha_scale) ------------
    cfgs = [(3, 8.86349, 0.16, 12000.), (4, 10., 0.4, 12000.), (21, 8.86349, 0.16, 12000.),
            (21, 8.64, 0., 1e4), (47, 8.86349, 0.16, 12000.), (128, 8.86349, 0.16, 12000.),
            (256, 8.86349, 0.16, 12000.)]
    for ci, (d, theta, shift, scale) in enumerate(cfgs):
        rs = np.random.RandomState(100 + ci)
        a = mfg_ac2.actor_critic(theta=theta, shift=shift, alpha_scale=scale, d=d)
        a.w = rs.rand(int((d + 1) * d / 2 + d + 1), 1)
        pis = test_pis(d, rs)
        if d >= 128:
            pis = pis[[1, 3]] if d == 128 else pis[[1]]
        rec = {k: [] for k in ('pi', 'alpha', 'alpha_deriv', 'gamma_seed', 'P', 'pi_next', 'reward',
                               'features', 'value', 'value_next', 'delta', 'gradient')}
        for k, p in enumerate(pis):
            seed = 1000 * ci + k
            np.random.seed(seed)
            Pm = a.sample_action(p)
            rec['pi'].append(p)
            rec['alpha'].append(a.mat_alpha.copy())
            rec['alpha_deriv'].append(a.mat_alpha_deriv.copy())
            rec['gamma_seed'].append(seed)
            rec['P'].append(Pm.copy())
            pn = np.transpose(Pm).dot(p)
            r = a.calc_reward(Pm, p, d)
            f = a.calc_features(p)
            fn = a.calc_features(pn)
            rec['pi_next'].append(pn)
            rec['reward'].append(r[0])
            rec['features'].append(f)
            rec['value'].append(a.calc_value(p)[0])
            rec['value_next'].append(a.calc_value(pn)[0])
            rec['delta'].append((r + 1 * fn.dot(a.w) - f.dot(a.w))[0])
            g = a.calc_gradient_vectorized(Pm.copy(), p)
            if d <= 21:
                gb = a.calc_gradient_basic(Pm.copy(), p)
                gl = a.calc_gradient(Pm.copy(), p)
                assert abs(gb - g) <= 1e-9 * max(1, abs(g)) and abs(gl - g) <= 1e-9 * max(1, abs(g))
            rec['gradient'].append(g)
        np.savez_compressed(
            os.path.join(OUT, 'functions_cfg%d_d%d.npz' % (ci, d)),
            d=d, theta=theta, shift=shift, alpha_scale=scale, w=a.w,
            **{k: np.array(v) for k, v in rec.items()})

    # ---- integer bookkeeping --------------------------------------------------------
    ints = {}
    a = mfg_ac2.actor_critic()
    rows = [[3, 9, 1, 9, 4, 0, 7], [10, 20, 30, 40, 50, 60, 70], [1, 2, 3, 4, 5, 6, 7]]
    ints['reorder_in'] = np.array(rows)
    ints['reorder_out'] = np.array(a.reorder([list(r) for r in rows]))
    ints['mat_pi0_d21'] = a.mat_pi0.copy()
    np.random.seed(7)
    ints['randint_seq'] = np.array([np.random.randint(a.num_start_samples) for _ in range(32)])
    import itertools
    for d in (3, 4, 21):
        ints['pairs_d%d' % d] = np.array(list(itertools.combinations_with_replacement(range(d), 2)))
    np.savez_compressed(os.path.join(OUT, 'ints_mfg_ac2.npz'), **ints)

    # ---- seeded train() traces ------------------------------------------------------
    for name, kw in (('c0_g1', dict(constant=0, gamma=1)), ('c1_g1', dict(constant=1, gamma=1)),
                     ('c0_g09', dict(constant=0, gamma=0.9)), ('c1_g09', dict(constant=1, gamma=0.9))):
        np.random.seed(2024)
        a = mfg_ac2.actor_critic()                   # consumes np.random.rand(F,1) for w
        w0 = a.w.copy()
        log = {'pi': [], 'theta_before': [], 'P': []}
        orig = a.sample_action

        def hooked(pi, _orig=orig, _a=a, _log=log):
            Pm = _orig(pi)
            _log['pi'].append(np.array(pi))
            _log['theta_before'].append(np.ravel(_a.theta)[0])
            _log['P'].append(Pm.copy())
            return Pm
        a.sample_action = hooked
        with quiet():
            a.train(num_episodes=6, lr_critic=0.1, lr_actor=0.001, consecutive=100, **kw)
        np.savez_compressed(
            os.path.join(OUT, 'train_mfg_ac2_%s.npz' % name), seed=2024, num_episodes=6,
            mat_pi0=a.mat_pi0, w0=w0, theta0=8.86349, shift=0.16, alpha_scale=12000., d=21,
            constant=kw['constant'], gamma=kw['gamma'], lr_critic=0.1, lr_actor=0.001,
            pi=np.array(log['pi']), theta_before=np.array(log['theta_before']), P=np.array(log['P']),
            theta_final=np.ravel(a.theta)[0], w_final=a.w)

    # ---- generate_trajectory ---------------------------------------------------------
    np.random.seed(5)
    a = mfg_ac2.actor_critic()
    np.random.seed(6)
    traj = a.generate_trajectory(a.mat_pi0[1], 16)
    np.savez_compressed(os.path.join(OUT, 'generate_trajectory_mfg_ac2.npz'), seed=6, pi0=a.mat_pi0[1],
                        theta=8.86349, shift=0.16, alpha_scale=12000., total_hours=16, traj=traj)


# --------------------------------------------------------------------------
def part_host_io(scratch):
    """Rows a12 / f2 / f4: the reference's own file bookkeeping and evaluation, run unmodified on
    synthetic files in its on-disk formats.  Stored: the input files' numbers, the directory order the reference saw
    (os.listdir order decides the np.random consumption order of evaluate), every returned value and every text line
    it wrote."""
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    os.chdir(scratch)
    d = 21
    write_start_states(scratch, d, 4, seed=0)
    import mfg_ac2
    out = {}

    # ---- normalize (mfg_ac2.py:116-137) -> init_pi0 (:179-208) round trip: raw count files with a header line ----
    rs = np.random.RandomState(31)
    os.makedirs('train_round2'); os.makedirs('norm_out')
    raw = rs.randint(0, 500, size=(3, 16, d + 3)).astype(np.float64)
    raw[:, :, 0] += 1                                      # no all-zero row
    for k in range(3):
        with open('train_round2/trend_distribution_day%d.csv' % (k + 1), 'w') as f:
            f.write(','.join('topic%d' % j for j in range(d + 3)) + '\n')
            for row in raw[k]:
                f.write(','.join('%d' % v for v in row) + '\n')
    a = mfg_ac2.actor_critic()
    a.normalize(indir='train_round2', outdir='norm_out', header=True)
    out['normalize_raw'] = raw
    out['normalize_text'] = np.array([read_text('norm_out/trend_distribution_day%d.csv' % (k + 1)) for k in range(3)])
    a.init_pi0(path_to_dir=os.getcwd() + '/norm_out')
    out['normalize_mat_pi0'] = a.mat_pi0.copy()

    # ---- evaluate (mfg_ac2.py:595-670) and gridsearch (:673-689) on a synthetic test set --------------------------
    rs = np.random.RandomState(32)
    os.makedirs('test_normalized_round2'); os.makedirs('eval_mfg_round2')
    emp = []
    for day in range(22, 26):
        m = rs.dirichlet(np.ones(d + 2), size=16)
        if day == 23:
            m[5, 3] = 0.0                                  # a zero entry: the 1e-100 rule of JSD (:556-557)
        np.savetxt('test_normalized_round2/trend_distribution_day%d.csv' % day, m, fmt='%.3e', delimiter=' ')
        emp.append(np.loadtxt('test_normalized_round2/trend_distribution_day%d.csv' % day, delimiter=' '))
    order = os.listdir(os.getcwd() + '/test_normalized_round2')
    out['eval_files'] = np.array(sorted(order))
    out['eval_listdir_order'] = np.array(order)
    out['eval_emp'] = np.array(emp)                        # [4,16,d+2], files in sorted-name order (day 22..25)
    np.random.seed(91)
    a = mfg_ac2.actor_critic()
    np.random.seed(92)
    res = a.evaluate(theta=8.86349, shift=0.5, alpha_scale=1e4, d=d, outfile='eval_mfg_round2/e.csv', write_header=1)
    out['eval_seed'] = 92
    out['eval_args'] = np.array([8.86349, 0.5, 1e4])
    out['eval_result'] = np.array(res)
    out['eval_csv'] = np.array(read_text('eval_mfg_round2/e.csv'))
    thetas, shifts, alphas = [7.0, 8.86349], [0.16, 0.5], [1e3, 1e4]
    np.random.seed(93)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        a.gridsearch(thetas, shifts, alphas, indir='test_normalized_round2', outfile='eval_mfg_round2/g.csv')
    best = eval(buf.getvalue().strip().split('\n')[-1], {'np': np, 'array': np.array, 'float64': np.float64})
    out['grid_seed'] = 93
    out['grid_thetas'] = np.array(thetas); out['grid_shifts'] = np.array(shifts); out['grid_alphas'] = np.array(alphas)
    out['grid_best'] = np.array([[float(v) for v in row] for row in best])     # printed argmin table (:689)
    out['grid_csv'] = np.array(read_text('eval_mfg_round2/g.csv'))

    # ---- train(write_file=1): the CSV log schema (mfg_ac2.py:441-445, :536-539) -----------------------------------
    os.makedirs('results')
    np.random.seed(2025)
    a = mfg_ac2.actor_critic()
    out['log_w0'] = a.w.copy()
    out['log_mat_pi0'] = a.mat_pi0.copy()
    with quiet():
        a.train(num_episodes=5, gamma=0.9, constant=0, consecutive=2, write_file=1)
    out['log_seed'] = 2025
    for name in ('theta', 'pi', 'reward'):
        out['log_' + name] = np.array(read_text('results/%s.csv' % name))
    out['log_theta_final'] = np.ravel(a.theta)[0]
    np.savez_compressed(os.path.join(OUT, 'host_io_mfg_ac2.npz'), **out)


def part_demonstrations(scratch):
    """f4: ac_irl.read_demonstrations / init_pi0_test / get_eval_transitions (ac_irl.py:164-219, :476-506) on files in
    the reference's layout.  read_demonstrations calls `pd.read_table(...).as_matrix()`; `pd` is only imported on
    Windows (ac_irl.py:16-22) and pandas 2.x removed DataFrame.as_matrix (it was an alias of `.values`), so the module
    global `pd` is bound to the REAL pandas with that one alias restored -- parsing is pandas' own."""
    import types
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    os.chdir(scratch)
    d, dim_action = 15, 20
    sys.modules['tensorflow'] = types.ModuleType('tensorflow')
    import pandas
    import ac_irl

    class _Frame:
        def __init__(self, df):
            self.df = df

        def as_matrix(self):
            return self.df.values

    pd_alias = types.SimpleNamespace(read_table=lambda *a, **k: _Frame(pandas.read_table(*a, **k)))
    ac_irl.pd = pd_alias
    rs = np.random.RandomState(41)
    os.makedirs('states'); os.makedirs('actions')
    states, actions = [], []
    for day in range(22, 25):
        st = rs.dirichlet(np.ones(dim_action), size=16)
        np.savetxt('states/trend_distribution_day%d.csv' % day, st, fmt='%.3e', delimiter=' ')
        with open('actions/action_day%d.txt' % day, 'w') as f:        # 15 blocks of dim_action rows, blank line after each
            for hour in range(15):
                blk = rs.dirichlet(np.ones(dim_action), size=dim_action)
                for row in blk:
                    row.tofile(f, sep=' ', format='%.3e')
                    f.write('\n')
                f.write('\n')
        states.append(read_text('states/trend_distribution_day%d.csv' % day))
        actions.append(read_text('actions/action_day%d.txt' % day))
    a = object.__new__(ac_irl.AC_IRL)
    a.d = d
    with quiet():
        demos = a.read_demonstrations('states', 'actions', dim_action, 22)
    a.init_pi0_test(path_to_dir=os.getcwd() + '/states', day_start=22)
    ev = a.get_eval_transitions(demos * 6)                              # 18 trajectories: index mod 15 wraps
    np.savez_compressed(os.path.join(OUT, 'demonstrations_ac_irl.npz'), d=d, dim_action=dim_action, start_day=22,
                        state_text=np.array(states), action_text=np.array(actions),
                        demo_pi=np.array([[p[0] for p in t] for t in demos]),
                        demo_P=np.array([[p[1] for p in t] for t in demos]),
                        mat_pi0_test=a.mat_pi0_test,
                        eval_pi=np.array([p[0] for p in ev]), eval_P=np.array([p[1] for p in ev]))


# --------------------------------------------------------------------------
def part_synthetic(scratch):
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    os.chdir(scratch)
    write_start_states(scratch, 21, 4, seed=0)
    import mfg_synthetic
    rs = np.random.RandomState(11)
    a = mfg_synthetic.actor_critic()
    pis, Ps, rs_out = [], [], []
    for k in range(4):
        p = rs.dirichlet(np.ones(21))
        Pm = rs.dirichlet(np.ones(21), size=21)
        pis.append(p); Ps.append(Pm)
        rs_out.append(float(np.ravel(a.calc_reward(Pm, p, 21))[0]))
    kat = float(np.ravel(a.calc_reward(np.array([[1, 3, 3], [4, 5, 6], [7, 8, 9]]),
                                       np.array([0.1, 0.2, 0.7]), 3))[0])
    np.savez_compressed(os.path.join(OUT, 'reward_mfg_synthetic.npz'), pi=np.array(pis), P=np.array(Ps),
                        reward=np.array(rs_out), kat_reward=kat)

    # backward value recursion V^n = r + P V^{n+1} and its two consistency metrics
    # (mfg_synthetic.py:741-812 evaluate_synthetic, :815-899 evaluate_synthetic_JSD)
    a = mfg_synthetic.actor_critic(theta=2.6, shift=0.0, alpha_scale=10000, d=21)
    # mfg_synthetic reads cwd/train_normalized/trend_distribution_day%d_reordered.csv (:181, :439)
    rs2 = np.random.RandomState(5)
    os.makedirs('train_normalized', exist_ok=True)
    for day in range(1, 5):
        np.savetxt('train_normalized/trend_distribution_day%d_reordered.csv' % day, rs2.dirichlet(np.ones(21), size=16),
                   fmt='%.3e', delimiter=' ')
    a.init_pi0(path_to_dir=os.getcwd() + '/train_normalized')
    captured = []
    orig = a.generate_trajectory

    def hooked(pi0, total_hours, _orig=orig):
        traj, acts = _orig(pi0, total_hours)
        captured.append((traj.copy(), acts.copy()))
        return traj, acts
    a.generate_trajectory = hooked
    np.random.seed(77)
    with quiet():
        l1_mean, l1_std = a.evaluate_synthetic(day_first=1, day_last=4)
    acts_l1 = np.array([c[1] for c in captured])
    captured.clear()
    np.random.seed(78)
    with quiet():
        js_mean, js_std = a.evaluate_synthetic_JSD(day_first=1, day_last=4)
    acts_js = np.array([c[1] for c in captured])
    np.savez_compressed(os.path.join(OUT, 'backward_value_mfg_synthetic.npz'), mat_pi0=a.mat_pi0, theta=2.6, shift=0.0,
                        alpha_scale=10000., actions_l1=acts_l1, l1_mean=l1_mean, l1_std=l1_std,
                        actions_jsd=acts_js, jsd_mean=js_mean, jsd_std=js_std,
                        reward_vector=a.calc_reward_vector(acts_l1[0, 0]))


# --------------------------------------------------------------------------
def fake_reward(pi, P):
    """Closed-form stand-in for the TF reward net (bounded like its tanh output)."""
    pi = np.asarray(pi, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    return np.tanh(5.0 * np.sum(pi * np.diagonal(P, axis1=-2, axis2=-1), axis=-1) - 0.3)


def part_ac_irl(scratch):
    import types
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    os.chdir(scratch)
    d = 21
    write_start_states(scratch, d, 4, seed=0)
    sys.modules['tensorflow'] = types.ModuleType('tensorflow')
    import ac_irl

    class FakeSess:
        def __init__(self, obj):
            self.obj = obj

        def run(self, fetch, feed_dict=None):
            pi = np.asarray(feed_dict['gen_states'])[0]
            Pm = np.asarray(feed_dict['gen_actions'])[0]
            return np.array([[fake_reward(pi, Pm)]])

    def make(theta, shift, scale, seed):
        np.random.seed(seed)
        a = object.__new__(ac_irl.AC_IRL)
        a.theta = theta; a.theta_initial = theta; a.shift = shift; a.alpha_scale = scale
        a.d = d
        a.w = a.init_w(d)
        a.init_pi0(path_to_dir=os.getcwd() + '/train_normalized_round2')
        a.num_start_samples = a.mat_pi0.shape[0]
        a.num_policies = 3
        a.list_policies = [theta] * a.num_policies
        a.reward_gen = 'reward_gen'; a.gen_states = 'gen_states'; a.gen_actions = 'gen_actions'
        a.sess = FakeSess(a)
        return a

    for name, kw in (('c0_g1', dict(constant=False, gamma=1, stop_criteria=-1)),
                     ('c1_g09', dict(constant=True, gamma=0.9, stop_criteria=-1)),
                     ('c0_g09_stop', dict(constant=False, gamma=0.9, stop_criteria=0.002))):
        a = make(8.64, 0.0, 1e4, 321)
        w0 = a.w.copy()
        log = {'pi': [], 'theta_before': [], 'P': []}
        orig = a.sample_action

        def hooked(pi, _orig=orig, _a=a, _log=log):
            Pm = _orig(pi)
            _log['pi'].append(np.array(pi))
            _log['theta_before'].append(float(np.ravel(_a.theta)[0]))
            _log['P'].append(Pm.copy())
            return Pm
        a.sample_action = hooked
        with quiet():
            a.train(max_episodes=5, lr_critic=0.1, lr_actor=0.001, consecutive=100, **kw)
        np.savez_compressed(
            os.path.join(OUT, 'train_ac_irl_%s.npz' % name), seed=321, max_episodes=5,
            mat_pi0=a.mat_pi0, w0=w0, theta0=8.64, shift=0.0, alpha_scale=1e4, d=d,
            constant=int(kw['constant']), gamma=kw['gamma'], stop_criteria=kw['stop_criteria'],
            lr_critic=0.1, lr_actor=0.001,
            pi=np.array(log['pi']), theta_before=np.array(log['theta_before']), P=np.array(log['P']),
            theta_final=float(np.ravel(a.theta)[0]), w_final=a.w,
            list_policies=np.array([float(np.ravel(t)[0]) for t in a.list_policies]),
            steps_run=len(log['P']))

    # generate_trajectories + ac_irl's calc_gradient_vectorized (recomputes alpha_deriv, stale alpha)
    a = make(8.64, 0.0, 1e4, 99)
    np.random.seed(17)
    with quiet():
        trajs = a.generate_trajectories(2)
    pis = np.array([[pair[0] for pair in t] for t in trajs])
    Ps = np.array([[pair[1] for pair in t] for t in trajs])
    grads = []
    for b in range(2):
        for t in range(15):
            a.sample_action  # alpha must belong to this pi: recompute like the reference would
            mat1 = np.repeat(pis[b, t].reshape(1, d), d, 0) - np.repeat(pis[b, t].reshape(d, 1), d, 1)
            a.mat_alpha = np.log(1 + np.exp(a.theta * (mat1 - a.shift)))
            grads.append(a.calc_gradient_vectorized(Ps[b, t].copy(), pis[b, t]))
    np.savez_compressed(os.path.join(OUT, 'generate_trajectories_ac_irl.npz'), seed=17, n=2,
                        mat_pi0=a.mat_pi0, theta=8.64, shift=0.0, alpha_scale=1e4,
                        pi=pis, P=Ps, gradient=np.array(grads).reshape(2, 15),
                        fake_reward=np.array([[fake_reward(pis[b, t], Ps[b, t]) for t in range(15)] for b in range(2)]))


# --------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--part', default=None)
    ap.add_argument('--scratch', default=None)
    args = ap.parse_args()
    if args.part is None:
        if not os.path.isdir(REF):
            sys.exit('reference not present at %s: fixtures can only be regenerated in the build container' % REF)
        os.makedirs(OUT, exist_ok=True)
        for part in ('mfg_ac2', 'host_io', 'demonstrations', 'synthetic', 'ac_irl'):
            with tempfile.TemporaryDirectory() as scratch:
                subprocess.run([sys.executable, os.path.abspath(__file__), '--part', part, '--scratch', scratch],
                               check=True)
        print('wrote', sorted(os.listdir(OUT)))
        return
    {'mfg_ac2': part_mfg_ac2, 'host_io': part_host_io, 'demonstrations': part_demonstrations,
     'synthetic': part_synthetic, 'ac_irl': part_ac_irl}[args.part](args.scratch)


if __name__ == '__main__':
    main()
