"""Reward network of the max-ent IRL experiments as a PyTorch module (runs on ROCm).

Restates the four TF-1.x graph builders of the reference's ``networks.py`` (r_net :13-43,
r_net_dropout_l1l2 :46-81, r_net_l1l2 :84-119, r_net_dropout :122-157):

    action [N,d,d] -> conv 5x5, 1 filter, SAME, ReLU -> conv 3x3, 2 filters, SAME, ReLU -> flatten 2 d^2
    -> FC n_fc3 ReLU (-> dropout keep 0.4) -> concat state [N,d] -> FC n_fc4 ReLU (-> dropout keep 0.4)
    -> FC 1, tanh                                                         => reward [N,1]

``tf.contrib.layers`` defaults are kept: Xavier-uniform weights, zero biases; the 'l1l2' variants add
``l1_l2_regularizer()`` (scale_l1 = scale_l2 = 1.0: sum|W| + sum W^2 / 2) on the fc3 / fc4 weights;
``tf.contrib.layers.dropout`` defaults to is_training=True, so in the reference dropout is ALSO active when
the net serves as the RL reward (ac_irl.py:683) -- reproduced by ``dropout_always=True`` (the default).
The flatten order is TF's NHWC (h, w, channel) so weights are layout compatible with a TF checkpoint.
Parity status: the reference holds no numeric test for this net and TF 1.x is not installable here, so the
module is checked against a NumPy restatement and hand-derived values only ("parity unpinned").
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as Fnn

REG_VARIANTS = ('none', 'dropout', 'l1l2', 'dropout_l1l2')


def conv2d_same_gemm(x, weight, bias):
    """Stride-1 SAME cross-correlation as im2col + matmul (x [N,C,H,W], weight [O,C,k,k]).
    Same math as F.conv2d(x, weight, bias, padding=k//2); written as unfold + GEMM so that on ROCm it runs on
    kernels already inside the PyTorch binary (rocBLAS) instead of MIOpen, whose first call per shape JIT
    compiles for tens of seconds on a fresh machine.  The maps here are tiny (d <= 64, <= 2 channels)."""
    N, C, H, W = x.shape
    O, _, k, _ = weight.shape
    cols = Fnn.unfold(x, kernel_size=k, padding=k // 2)                  # [N, C*k*k, H*W]
    out = torch.matmul(weight.reshape(O, C * k * k), cols)               # [N, O, H*W]
    return (out + bias.reshape(1, O, 1)).reshape(N, O, H, W)


class RewardNet(nn.Module):

    def __init__(self, d=15, reg='dropout_l1l2', f1=1, k1=5, f2=2, k2=3, n_fc3=8, n_fc4=4, keep_prob=0.4,
                 dropout_always=True):
        super().__init__()
        if reg not in REG_VARIANTS:
            raise ValueError('reg must be one of %s' % (REG_VARIANTS,))
        self.d, self.reg, self.f2 = d, reg, f2
        self.use_dropout = 'dropout' in reg
        self.use_l1l2 = 'l1l2' in reg
        self.keep_prob = keep_prob
        self.dropout_always = dropout_always
        self.conv1 = nn.Conv2d(1, f1, k1, stride=1, padding=k1 // 2)
        self.conv2 = nn.Conv2d(f1, f2, k2, stride=1, padding=k2 // 2)
        self.fc3 = nn.Linear(f2 * d * d, n_fc3)
        self.fc4 = nn.Linear(n_fc3 + d, n_fc4)
        self.out = nn.Linear(n_fc4, 1)
        for m in (self.conv1, self.conv2, self.fc3, self.fc4, self.out):
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)

    def _drop(self, x):
        if not self.use_dropout:
            return x
        return Fnn.dropout(x, p=1.0 - self.keep_prob, training=self.dropout_always or self.training)

    def forward(self, state, action):
        """state [N,d], action [N,d,d] -> reward [N,1] in (-1, 1)."""
        d = self.d
        x = action.reshape(-1, 1, d, d)
        s = state.reshape(-1, d)
        x = Fnn.relu(conv2d_same_gemm(x, self.conv1.weight, self.conv1.bias))
        x = Fnn.relu(conv2d_same_gemm(x, self.conv2.weight, self.conv2.bias))
        x = x.permute(0, 2, 3, 1).reshape(-1, self.f2 * d * d)     # NHWC flatten, networks.py:67
        x = self._drop(Fnn.relu(self.fc3(x)))
        x = torch.cat([x, s], dim=1)                                # networks.py:72
        x = self._drop(Fnn.relu(self.fc4(x)))
        return torch.tanh(self.out(x))

    # TF checkpoint layout (tf.contrib.layers variables under the reference's scopes, networks.py:62-79 inside
    # tf.variable_scope("reward"), ac_irl.py:246): conv kernels HWIO, dense kernels [in, out]
    TF_NAMES = (('conv1', 'reward/conv1'), ('conv2', 'reward/conv2'), ('fc3', 'reward/fc3'), ('fc4', 'reward/fc4'),
                ('out', 'reward/out'))

    def tf_variables(self):
        """This module's parameters as {TF variable name: float32 ndarray in TF's layout}: '<scope>/weights' is
        [kh, kw, in, out] (HWIO) for the convolutions and [in, out] for the dense layers, '<scope>/biases' is [out] --
        the tensors `tf.train.Saver` writes for the reference's graph (ac_irl.py:948, log/model_<reg>_<n3>_<n4>.ckpt)."""
        out = {}
        for attr, scope in self.TF_NAMES:
            m = getattr(self, attr)
            w = m.weight.detach().cpu().float()
            out[scope + '/weights'] = (w.permute(2, 3, 1, 0) if w.dim() == 4 else w.t()).contiguous().numpy()
            out[scope + '/biases'] = m.bias.detach().cpu().float().numpy().copy()
        return out

    def load_tf_variables(self, variables):
        """Inverse of tf_variables(): load {name: array} read from a TF-1.x checkpoint of the reference (e.g.
        `r = tf.train.load_checkpoint(path); {n: r.get_tensor(n) for n in names}` on a machine that has TensorFlow) into
        this module.  HWIO -> OIHW for the convolutions, [in, out] -> [out, in] for the dense layers; the fc3 input order
        needs no permutation because forward() flattens NHWC like networks.py:67.  Shapes are checked."""
        import numpy as np
        with torch.no_grad():
            for attr, scope in self.TF_NAMES:
                m = getattr(self, attr)
                w = torch.as_tensor(np.asarray(variables[scope + '/weights'], dtype=np.float32))
                w = w.permute(3, 2, 0, 1) if w.dim() == 4 else w.t()
                b = torch.as_tensor(np.asarray(variables[scope + '/biases'], dtype=np.float32)).reshape(-1)
                if tuple(w.shape) != tuple(m.weight.shape) or tuple(b.shape) != tuple(m.bias.shape):
                    raise ValueError('%s: checkpoint shape %s / %s does not fit this network (%s / %s)'
                                     % (scope, tuple(w.shape), tuple(b.shape), tuple(m.weight.shape), tuple(m.bias.shape)))
                m.weight.copy_(w.to(m.weight.dtype))
                m.bias.copy_(b.to(m.bias.dtype))
        return self

    def regularization(self):
        """Sum of the l1_l2 penalties collected in tf.GraphKeys.REGULARIZATION_LOSSES (ac_irl.py:409-411)."""
        if not self.use_l1l2:
            return self.fc3.weight.new_zeros(())
        reg = 0.0
        for m in (self.fc3, self.fc4):
            reg = reg + m.weight.abs().sum() + 0.5 * (m.weight ** 2).sum()
        return reg


def maxent_irl_loss(reward_demo, reward_gen, num_demo_samples, num_sampled_trajectories, reg_loss=None, steps=15):
    """Guided-cost-learning loss of ac_irl.py:390-413 (importance weights z disabled there, :404-406):
        -(1/N_demo) sum r_demo  +  log( (1/M) sum_traj exp( sum_t r_gen ) )  [+ sum reg].
    Returns (loss, first_term, second_term)."""
    first = -1.0 / num_demo_samples * reward_demo.sum()
    per_traj = reward_gen.reshape(num_sampled_trajectories, steps).sum(dim=1)
    second = torch.log(1.0 / num_sampled_trajectories * torch.exp(per_traj).sum())
    loss = first + second
    if reg_loss is not None:
        loss = loss + reg_loss
    return loss, first, second
