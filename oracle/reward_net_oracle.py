"""NumPy restatement of the reference's reward network forward pass and IRL loss.  TEST INFRASTRUCTURE.

PARITY UNPINNED: ``networks.py`` / ``ac_irl.py:382-427`` need TensorFlow 1.x (``tf.contrib``), which cannot be
imported or installed here, and the reference holds no numeric test for the net.  This file therefore
follows the published semantics of the TF ops named in the reference (networks.py:46-81):
conv2d SAME stride 1 (cross-correlation, zero padding), ReLU, NHWC flatten, fully_connected, tanh;
``l1_l2_regularizer()`` = sum|W| + sum W^2 / 2 (scale_l1 = scale_l2 = 1); loss of ac_irl.py:390-413.
It is used to check the PyTorch module (weights shared), with dropout disabled.

Op semantics relied on (TensorFlow 1.x API documentation; none of it can be executed here):
  * tf.nn.conv2d / tf.contrib.layers.conv2d, NHWC, filter [kh, kw, in, out]: "computes ... output[b, i, j, k] =
    sum_{di, dj, q} input[b, strides[1]*i + di, strides[2]*j + dj, q] * filter[di, dj, q, k]" -- a CROSS-CORRELATION (no
    kernel flip); padding "SAME", stride 1, odd k: out size = in size, pad_total = k - 1 split as (k-1)/2 before and
    after, zeros.  => scipy.signal.correlate2d(x, w, mode='same') per (in, out) channel pair, summed over `in`
    (checked in tests/test_reward_net.py::test_oracle_conv_is_tf_same_cross_correlation with asymmetric kernels).
  * tf.reshape(conv2, [-1, f2*d*d]) of an NHWC tensor (networks.py:67): row-major over (h, w, c): index (h*d + w)*f2 + c.
  * tf.contrib.layers.fully_connected: activation(x . W + b), W [in, out]; weights_initializer = xavier_initializer()
    (uniform, limit sqrt(6 / (fan_in + fan_out))), biases_initializer = zeros.
  * tf.contrib.layers.l1_l2_regularizer(scale_l1=1.0, scale_l2=1.0) = l1_regularizer + l2_regularizer with
    l1 = scale_l1 * sum |W|, l2 = scale_l2 * tf.nn.l2_loss(W), and tf.nn.l2_loss(t) = sum(t ** 2) / 2.
  * tf.contrib.layers.dropout(x, keep_prob, is_training=True): tf.nn.dropout -- keeps each unit with probability
    keep_prob and scales the kept ones by 1 / keep_prob (inverted dropout); is_training defaults to True, so the
    reference applies it also when the net serves as the RL reward (ac_irl.py:683).
  * Checkpoint variables (tf.train.Saver, ac_irl.py:948): reward/<scope>/weights, reward/<scope>/biases with the layouts
    above; discrete_mean_field_game_amd.networks.RewardNet.load_tf_variables / tf_variables convert to / from them.
"""
import numpy as np


def conv2d_same(x, w, b):
    """x [N,H,W,Cin], w [kh,kw,Cin,Cout] (TF layout), stride 1, SAME zero padding."""
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((N, H + 2 * ph, W + 2 * pw, Cin))
    xp[:, ph:ph + H, pw:pw + W] = x
    out = np.zeros((N, H, W, Cout))
    for u in range(kh):
        for v in range(kw):
            out += np.einsum('nhwc,co->nhwo', xp[:, u:u + H, v:v + W], w[u, v])
    return out + b


def dropout_masks(keep_prob, seed, sample_offset, N, n3, n4):
    """The dropout masks the HIP kernel draws (csrc/mfg_reward_net.hip): unit o of FC3 / FC4 of sample n is kept iff
    u01(Philox4x32-10(key = seed, counter = (o, 3 | 4, sample_offset + n, block 0)).x) <= keep_prob; kept units are scaled
    by 1 / keep_prob (tf.nn.dropout).  TF's own generator cannot be matched; what this pins is that the kernel evaluates
    the documented function of the documented bits, so a run WITH dropout (the reference's default reg='dropout_l1l2',
    active also when the net serves as the RL reward) can be replayed.  Returns (m3 [N,n3], m4 [N,n4]) of 0 / (1/keep)."""
    from oracle import philox_ref as PR
    n = np.arange(N, dtype=np.uint64) + np.uint64(sample_offset)
    inv = np.float32(1.0) / np.float32(keep_prob)
    out = []
    for units, step in ((n3, 3), (n4, 4)):
        o = np.arange(units, dtype=np.uint64)
        r = PR.philox_elem(int(seed), o[None, :], step, n[:, None], 0)[0]
        u = PR.u01(r)
        out.append(np.where(u <= np.float32(keep_prob), inv, np.float32(0.0)).astype(np.float64))
    return out[0], out[1]


def forward(params, state, action, dropout=None):
    """params: dict of TF-layout arrays conv1_w [5,5,1,1], conv1_b, conv2_w [3,3,1,2], conv2_b,
    fc3_w [2d^2,n3], fc3_b, fc4_w [n3+d,n4], fc4_b, out_w [n4,1], out_b.  Returns [N,1].
    dropout = (keep_prob, seed, sample_offset): apply the kernel's counter-based masks after fc3 and fc4."""
    N, d = state.shape
    x = action.reshape(N, d, d, 1).astype(np.float64)
    x = np.maximum(conv2d_same(x, params['conv1_w'], params['conv1_b']), 0)
    x = np.maximum(conv2d_same(x, params['conv2_w'], params['conv2_b']), 0)
    x = x.reshape(N, -1)                                            # NHWC flatten (networks.py:67)
    x = np.maximum(x.dot(params['fc3_w']) + params['fc3_b'], 0)
    m3 = m4 = None
    if dropout is not None:
        m3, m4 = dropout_masks(dropout[0], dropout[1], dropout[2], N, params['fc3_w'].shape[1], params['fc4_w'].shape[1])
        x = x * m3                                                   # networks.py:70
    x = np.concatenate([x, state.astype(np.float64)], axis=1)        # networks.py:72
    x = np.maximum(x.dot(params['fc4_w']) + params['fc4_b'], 0)
    if m4 is not None:
        x = x * m4                                                   # networks.py:75
    return np.tanh(x.dot(params['out_w']) + params['out_b'])


def l1_l2(params):
    return sum(np.abs(params[k]).sum() + 0.5 * (params[k] ** 2).sum() for k in ('fc3_w', 'fc4_w'))


def irl_loss(r_demo, r_gen, n_demo, n_traj, reg=0.0, steps=15):
    first = -1.0 / n_demo * np.sum(r_demo)
    second = np.log(1.0 / n_traj * np.sum(np.exp(np.reshape(r_gen, (n_traj, steps)).sum(1))))
    return first + second + reg, first, second


def params_from_torch(net):
    """Convert a networks.RewardNet state to the TF layouts used above."""
    g = lambda t: t.detach().cpu().double().numpy()
    return {
        'conv1_w': g(net.conv1.weight).transpose(2, 3, 1, 0), 'conv1_b': g(net.conv1.bias),
        'conv2_w': g(net.conv2.weight).transpose(2, 3, 1, 0), 'conv2_b': g(net.conv2.bias),
        'fc3_w': g(net.fc3.weight).T, 'fc3_b': g(net.fc3.bias),
        'fc4_w': g(net.fc4.weight).T, 'fc4_b': g(net.fc4.bias),
        'out_w': g(net.out.weight).T, 'out_b': g(net.out.bias),
    }


# ---------------------------------------------------------------------------------------------------------------------
# Training step (ac_irl.py:382-418 loss + tf.train.AdamOptimizer, :804-846 update_reward), fp64.  PARITY UNPINNED like the
# forward pass: this is the hand-derived gradient of the restated graph, checked against PyTorch autograd of
# networks.RewardNet and against central differences in tests/test_reward_net.py.
# ---------------------------------------------------------------------------------------------------------------------
def forward_cache(params, state, action, masks=None):
    """forward() that also returns what backward() needs.  masks = (m3 [N,n3], m4 [N,n4]) of 0 / (1/keep) or None."""
    N, d = state.shape
    x0 = action.reshape(N, d, d, 1).astype(np.float64)
    z1 = conv2d_same(x0, params['conv1_w'], params['conv1_b'])
    a1 = np.maximum(z1, 0)
    z2 = conv2d_same(a1, params['conv2_w'], params['conv2_b'])
    a2 = np.maximum(z2, 0)
    flat = a2.reshape(N, -1)
    z3 = flat.dot(params['fc3_w']) + params['fc3_b']
    h3 = np.maximum(z3, 0)
    if masks is not None:
        h3 = h3 * masks[0]
    in4 = np.concatenate([h3, state.astype(np.float64)], axis=1)
    z4 = in4.dot(params['fc4_w']) + params['fc4_b']
    h4 = np.maximum(z4, 0)
    if masks is not None:
        h4 = h4 * masks[1]
    r = np.tanh(h4.dot(params['out_w']) + params['out_b'])
    # smallest |pre-activation| of a ReLU, relative to the magnitude of the terms it sums: an fp32 evaluation may land on the
    # other side of a kink this close to zero, and the gradient then differs by that unit's whole contribution -- a dense
    # unit carries a sample's whole share of the fc gradients, a conv pixel ~1e-4 of a conv-weight gradient (one of ~8 000
    # pixel terms).  The randomised soak (tools/rn_train_soak.py) skips such draws.
    m3 = np.abs(flat).dot(np.abs(params['fc3_w'])) + np.abs(params['fc3_b'])
    m4 = np.abs(in4).dot(np.abs(params['fc4_w'])) + np.abs(params['fc4_b'])
    m1 = conv2d_same(np.abs(x0), np.abs(params['conv1_w']), np.abs(params['conv1_b']))
    m2 = conv2d_same(np.abs(a1), np.abs(params['conv2_w']), np.abs(params['conv2_b']))
    kink = min(float((np.abs(z) / np.maximum(m, 1e-300)).min()) for z, m in ((z1, m1), (z2, m2), (z3, m3), (z4, m4)))
    return r, dict(x0=x0, a1=a1, a2=a2, flat=flat, h3=h3, in4=in4, h4=h4, r=r, masks=masks, kink=kink)


def _conv_backward(x, w, dz):
    """Gradients of z = conv2d_same(x, w, b) given dz [N,H,W,Cout]: (dx, dw, db)."""
    N, H, W, Cin = x.shape
    kh, kw, _, Cout = w.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((N, H + 2 * ph, W + 2 * pw, Cin))
    xp[:, ph:ph + H, pw:pw + W] = x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w, dtype=np.float64)
    for u in range(kh):
        for v in range(kw):
            dw[u, v] = np.einsum('nhwc,nhwo->co', xp[:, u:u + H, v:v + W], dz)
            dxp[:, u:u + H, v:v + W] += np.einsum('nhwo,co->nhwc', dz, w[u, v])
    return dxp[:, ph:ph + H, pw:pw + W], dw, dz.sum(axis=(0, 1, 2))


def backward(params, cache, dr):
    """d (sum_n dr_n r_n) / d params for dr [N,1]; TF layouts like `params`."""
    m = cache['masks']
    dzo = dr * (1.0 - cache['r'] ** 2)
    g = {'out_w': cache['h4'].T.dot(dzo), 'out_b': dzo.sum(0)}
    dh4 = dzo.dot(params['out_w'].T)
    dz4 = dh4 * (cache['h4'] > 0) * (m[1] if m is not None else 1.0)
    g['fc4_w'] = cache['in4'].T.dot(dz4)
    g['fc4_b'] = dz4.sum(0)
    n3 = params['fc3_w'].shape[1]
    dh3 = dz4.dot(params['fc4_w'].T)[:, :n3]
    dz3 = dh3 * (cache['h3'] > 0) * (m[0] if m is not None else 1.0)
    g['fc3_w'] = cache['flat'].T.dot(dz3)
    g['fc3_b'] = dz3.sum(0)
    da2 = dz3.dot(params['fc3_w'].T).reshape(cache['a2'].shape)
    dz2 = da2 * (cache['a2'] > 0)
    da1, g['conv2_w'], g['conv2_b'] = _conv_backward(cache['a1'], params['conv2_w'], dz2)
    dz1 = da1 * (cache['a1'] > 0)
    _, g['conv1_w'], g['conv1_b'] = _conv_backward(cache['x0'], params['conv1_w'], dz1)
    return g


def irl_loss_and_grad(params, demo_state, demo_action, gen_state, gen_action, n_demo_div, n_traj, l1l2=False, steps=15,
                      masks=None):
    """Loss of ac_irl.py:390-413 and its gradient for one update_reward batch.  masks: (m3, m4) over the concatenated batch
    (demonstrations first) or None.  Returns ((loss, first, second, reg), grads dict, rewards [N,1])."""
    nd = demo_state.shape[0]
    state = np.concatenate([demo_state, gen_state], 0)
    action = np.concatenate([demo_action, gen_action], 0)
    r, cache = forward_cache(params, state, action, masks)
    reg = l1_l2(params) if l1l2 else 0.0
    loss, first, second = irl_loss(r[:nd], r[nd:], n_demo_div, n_traj, reg, steps)
    S = r[nd:].reshape(n_traj, steps).sum(1)
    soft = np.exp(S - S.max())
    soft = soft / soft.sum()
    dr = np.concatenate([np.full((nd, 1), -1.0 / n_demo_div), np.repeat(soft, steps)[:, None]], 0)
    g = backward(params, cache, dr)
    if l1l2:
        for k in ('fc3_w', 'fc4_w'):
            g[k] = g[k] + np.sign(params[k]) + params[k]
    return (loss, first, second, reg), g, r


def adam_tf(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (ac_irl.py:417): lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t); m, v exponential averages;
    p -= lr_t m / (sqrt(v) + eps).  Returns (p, m, v)."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    return p - lr_t * m / (np.sqrt(v) + eps), m, v


FLAT_ORDER = ('conv1_w', 'conv1_b', 'conv2_w', 'conv2_b', 'fc3_w', 'fc3_b', 'fc4_w', 'fc4_b', 'out_w', 'out_b')


def flatten_like_kernel(tf_arrays):
    """TF-layout dict -> the flat parameter order of mfg_reward_net_train_step (PyTorch layouts: conv OIHW, dense [out, in])."""
    out = []
    for k in FLAT_ORDER:
        a = np.asarray(tf_arrays[k], dtype=np.float64)
        if a.ndim == 4:
            a = a.transpose(3, 2, 0, 1)
        elif a.ndim == 2:
            a = a.T
        out.append(a.reshape(-1))
    return np.concatenate(out)
