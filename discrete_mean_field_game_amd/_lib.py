"""ctypes binding of the C ABI in include/mfg_hip.h (libmfg_hip.so, built in-tree by csrc/Makefile).

There is deliberately NO fallback: if the HIP library is missing or a call fails, an exception is
raised.  The product path never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('MFG_HIP_LIB') or os.path.join(CSRC, 'libmfg_hip.so')   # MFG_HIP_LIB: alternative build

MFG_MAX_D = 512
REWARD_MFG_AC2, REWARD_SYNTHETIC, REWARD_EXTERNAL = 0, 1, 2
ROLLOUT_WRITE_P, ROLLOUT_TD, ROLLOUT_DISCOUNT_POW, ROLLOUT_F64, TRAIN_APPLY = 1, 2, 4, 8, 16
PRECISION_F64, PRECISION_MIXED = 0, 1
STATUS_MIXED_RANGE = 1
ECOMM = -6                      # MFG_ECOMM: the call aborted its RCCL communicator, the handle is dead
RN_TRAIN_MAX_TRAJ = 64          # MFG_RN_TRAIN_MAX_TRAJ
PRECISIONS = {'f64': PRECISION_F64, 'mixed': PRECISION_MIXED, 0: 0, 1: 1}


class MfgError(RuntimeError):
    code = 0                    # the negative MFG_E* code of the failed call (0 for errors raised on the Python side)


def build(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ['make', '-C', CSRC]
    if force:
        args.append('-B')
    subprocess.run(args, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB_PATH


_p = C.c_void_p
_i64, _i32, _u64, _u32, _f64, _sz = C.c_int64, C.c_int, C.c_uint64, C.c_uint32, C.c_double, C.c_size_t

# symbol -> (restype, argtypes); mirrors include/mfg_hip.h one to one
SIGNATURES = {
    'mfg_last_error': (C.c_char_p, []),
    'mfg_abi_version': (_i32, []),
    'mfg_set_core_mapping': (_i32, [_i32]),
    'mfg_init': (_i32, []),
    'mfg_status': (_i32, [C.POINTER(C.c_uint)]),
    'mfg_clear_status': (_i32, []),
    'mfg_ctx_create': (_i32, [C.POINTER(C.c_void_p)]),
    'mfg_ctx_destroy': (_i32, [_p]),
    'mfg_ctx_bind': (_i32, [_p]),
    'mfg_ctx_current': (_p, []),
    'mfg_ctx_status': (_i32, [_p, C.POINTER(C.c_uint)]),
    'mfg_ctx_clear_status': (_i32, [_p]),
    'mfg_ctx_adopt_comm': (_i32, [_p, _p]),
    'mfg_ctx_comm': (_p, [_p]),
    'mfg_device_info': (_i32, [C.POINTER(C.c_int), C.c_char_p, _i32]),
    'mfg_feature_index': (_i64, [_i32, _i32, _i32]),
    'mfg_num_features': (_i64, [_i32]),
    'mfg_workspace_bytes': (_sz, [_i64, _i32]),
    'mfg_gather_start': (_i32, [_p, _i64, _p, _i64, _i32, _p, _p]),
    'mfg_draw_start': (_i32, [_p, _i64, _i64, _i32, _u64, _u32, _u64, _p, _p, _p]),
    'mfg_alpha': (_i32, [_p, _i64, _i32, _p, _f64, _p, _p, _p]),
    'mfg_dirichlet_from_gamma': (_i32, [_p, _i64, _i32, _p, _p]),
    'mfg_sample_dirichlet': (_i32, [_p, _i64, _i32, _p, _f64, _f64, _u64, _u32, _u64, _i32, _p, _p]),
    'mfg_philox_raw': (_i32, [_u64, _u32, _u32, _u32, _u32, _i64, _p, _p]),
    'mfg_step_given_P': (_i32, [_p, _p, _i64, _i32, _i32, _p, _p, _p]),
    'mfg_value': (_i32, [_p, _p, _i64, _i32, _p, _p]),
    'mfg_features': (_i32, [_p, _i64, _i32, _p, _p]),
    'mfg_score': (_i32, [_p, _p, _i64, _i32, _p, _f64, _i32, _p, _p]),
    'mfg_td_pg_accumulate': (_i32, [_p, _p, _p, _p, _p, _p, _f64, _f64, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _sz,
                                   _p]),
    'mfg_apply_update': (_i32, [_p, _i32, _f64, _f64, _p, _p, _p, _p]),
    'mfg_rollout': (_i32, [_p, _i64, _i32, _i32, _p, _f64, _f64, _p, _f64, _i32, _u64, _u32, _u64, _i32,
                           _p, _p, _p, _p, _p, _p, _p, _i32, _p, _sz, _p]),
    'mfg_jsd': (_i32, [_p, _p, _i64, _i32, _p, _p]),
    'mfg_grad_accumulate': (_i32, [_p, _i64, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _i32, _p, _sz, _p]),
    'mfg_grad_apply': (_i32, [_p, _i64, _p, _p, _p, _i64, _i32, _i32, _i32, _p, _f64, _f64, _p, _p, _p, _p, _sz, _p]),
    'mfg_train_episode': (_i32, [_p, _p, _i64, _i32, _i32, _p, _f64, _f64, _p, _f64, _i32, _u64, _u32, _u64, _i32, _f64, _f64,
                                 _p, _p, _p, _p, _p, _p, _sz, _p]),
    'mfg_train_rollout': (_i32, [_p, _i64, _p, _i64, _i32, _i32, _p, _f64, _f64, _p, _f64, _i32, _u64, _u32, _u64, _i32, _f64,
                                 _f64, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'mfg_train_rollouts': (_i32, [_p, _i64, _i64, _i32, _i32, _i64, _i64, _i32, _p, _f64, _f64, _p, _f64, _i32, _u64, _u32, _u64,
                                  _i32, _f64, _f64, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'mfg_train_rollout_deferred': (_i32, [_p, _i64, _p, _i64, _i32, _i32, _p, _p, _p, _f64, _f64, _p, _p, _p, _f64, _f64, _f64,
                                          _i32, _u64, _u32, _u64, _i32, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'mfg_train_episodes': (_i32, [_p, _i64, _p, _p, _i64, _i32, _i32, _i64, _i64, _i32, _p, _f64, _f64, _p, _f64, _i32, _u64, _u32,
                                  _u64, _i32, _f64, _f64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'mfg_dist_available': (_i32, []),
    'mfg_dist_unique_id': (_i32, [_p]),
    'mfg_dist_init': (_i32, [_p, _i32, _i32, C.POINTER(C.c_void_p)]),
    'mfg_dist_destroy': (_i32, [_p]),
    'mfg_dist_abort': (_i32, [_p]),
    'mfg_dist_all_reduce': (_i32, [_p, _p, _i64, _p]),
    'mfg_train_rollouts_dist': (_i32, [_p, _p, _i64, _i64, _i32, _i32, _i64, _i64, _i32, _p, _p, _p, _p, _f64, _f64, _f64, _i32, _u64,
                                       _u32, _u64, _i32, _f64, _f64, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'mfg_policy_logpdf': (_i32, [_p, _p, _i64, _i32, _p, _i32, _f64, _f64, _f64, _f64, _p, _p]),
    'mfg_backward_value': (_i32, [_p, _i64, _i32, _i32, _p, _p, _p, _p]),
    'mfg_reward_net_forward': (_i32, [_p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                     _p, C.c_float, _u64, _u64, _p, _p]),
    'mfg_reward_net_num_params': (_i64, [_i32, _i32, _i32, _i32, _i32, _i32]),
    'mfg_reward_net_param_offsets': (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_int64)]),
    'mfg_reward_net_train_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32, _i32, _i32, _i64]),
    'mfg_reward_net_train_step': (_i32, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, C.POINTER(C.c_int32), _i32, _p, _p,
                                        C.POINTER(C.c_int32), _i32, _i32, _i32, C.c_float, _i32, _u64, _f64, _f64, _f64, _f64, _i64,
                                        _i32, _p, _p, _p, _sz, _p]),
    'mfg_reward_net_adam': (_i32, [_p, _p, _p, _p, _i64, _f64, _f64, _f64, _f64, _i64, _p]),
}



class RewardNetStruct(C.Structure):
    """mfg_reward_net_t of include/mfg_hip.h."""
    _fields_ = ([(n, C.c_int) for n in ('k1', 'f2', 'k2', 'n3', 'n4')]
                + [(n, C.c_void_p) for n in ('conv1_w', 'conv1_b', 'conv2_w', 'conv2_b', 'fc3_w', 'fc3_b', 'fc4_w', 'fc4_b',
                                             'out_w', 'out_b')]
                + [('keep_prob', C.c_float)])


SIGNATURES['mfg_train_episode_irl'] = (_i32, [_p, _p, _i64, _i32, _i32, _p, _f64, _f64, _p, _f64, _u64, _u32, _u64, _i32, _f64,
                                              _f64, C.POINTER(RewardNetStruct), _u64, _u64, _u64, _p, _p, _p, _p, _p, _p, _p,
                                              _sz, _p])

SIGNATURES['mfg_train_episode_irl_draw'] = (_i32, [_p, _i64] + SIGNATURES['mfg_train_episode_irl'][1])

SIGNATURES['mfg_train_rollout_irl'] = (_i32, [_p, _i64, _p, _i64, _i32, _i32, _p, _f64, _f64, _p, _f64, _u64, _u32, _u64, _i32, _f64,
                                              _f64, C.POINTER(RewardNetStruct), _u64, _u64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz,
                                              _p])

_lib = None


def lib():
    """Load libmfg_hip.so (once) and bind every symbol.  Raises MfgError when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfgError('HIP extension not built: %s is missing (run __graft_entry__.build() / make -C %s)'
                           % (LIB_PATH, CSRC))
        # PyTorch-ROCm brings its own libamdhip64; whichever copy is loaded FIRST owns the process.  Loading this library
        # before torch binds it to /opt/rocm's runtime, torch then loads a second one, and kernels / symbols registered
        # with one runtime are unknown to the other ("h(z) table initialisation failed").  So torch goes first.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        err = MfgError('%s failed (%d): %s' % (what, rc, lib().mfg_last_error().decode()))
        err.code = int(rc)
        raise err
