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
