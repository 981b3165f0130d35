"""Multi-GPU layout of the hot path: the trajectory batch shards across ranks, one all-reduce per update.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
Trajectories are independent given (theta, w), so no P or pi ever crosses GPUs; the only exchange is the
fused gradient buffer  G = [G_w (F) | G_theta | sum_reward | count]  (fp64, F+3 entries, 2 KB at d=21,
265 KB at d=256): latency bound, so it is sent as ONE collective per update.
Every rank then applies the identical update, so (theta, w) stay replicated without a broadcast.
The RNG is counter based and keyed by the GLOBAL trajectory id (traj_offset + b), so results do not depend
on the world size.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    global_batch: int
    local_batch: int
    traj_offset: int


def shard_batch(global_batch: int, rank: int, world: int) -> Shard:
    """Contiguous split of [0, global_batch) into `world` shards; the first (global_batch % world) ranks
    get one extra trajectory.  Integer bookkeeping: shards tile the range exactly once."""
    if world < 1 or not (0 <= rank < world) or global_batch < 0:
        raise ValueError('bad shard request')
    base, extra = divmod(global_batch, world)
    local = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return Shard(rank, world, global_batch, local, offset)


def current_shard(global_batch: int, group=None) -> Shard:
    if dist.is_available() and dist.is_initialized():
        return shard_batch(global_batch, dist.get_rank(group), dist.get_world_size(group))
    return shard_batch(global_batch, 0, 1)


def all_reduce_gradients_(G: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """In-place SUM of the fused gradient buffer over all ranks (no-op for a single process).
    The count entry G[F+2] is summed too, so dividing by it afterwards gives the global batch mean.
    force: issue the collective also on a 1-rank communicator (bench.py --force-dist: the RCCL call of the multi-GPU
    update executed on a 1-GPU box)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        if G.is_cuda and dist.get_backend(group) != 'nccl':
            # gloo (CPU tests, several ranks sharing one GPU): no device collectives on this build, stage the 2 KB
            # buffer through the host; production runs use RCCL ("nccl") on the device tensor directly
            h = G.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            G.copy_(h)
        else:
            dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
    return G


def broadcast_start_indices(idx, group=None, device=None):
    """Rank 0's start-state index vector for every rank (int64 numpy in, int64 numpy out).  A few KB once per episode;
    goes through a device tensor under RCCL ("nccl" has no host tensors) and through the host under gloo."""
    import numpy as np
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return idx
    on_dev = dist.get_backend(group) == 'nccl'
    t = torch.as_tensor(np.ascontiguousarray(idx, dtype=np.int64), device=device if on_dev else 'cpu')
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return t.cpu().numpy()


def all_reduce_mean_flat_(tensors, group=None):
    """Average a list of tensors over the ranks with ONE collective: flatten, all-reduce (SUM), divide by the world size,
    scatter back in place.  Used for the replicated reward network's gradient (7 235 floats at d = 21,
    n_fc3 = 8, n_fc4 = 4 -- ten parameter tensors, one message).  No-op for a single process.  Returns the number of
    collectives issued (0 or 1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 or not tensors:
        return 0
    flat = torch.cat([t.reshape(-1) for t in tensors])
    if flat.is_cuda and dist.get_backend(group) != 'nccl':             # gloo (tests): staged through the host
        h = flat.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(h)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return 1


def broadcast_seed(value: int, group=None, device=None) -> int:
    """Rank 0's `value` (a non-negative integer below 2^62) on every rank: the ranks of a job re-seed their HOST samplers
    from it (Python `random` / torch generator of AC_IRL.update_reward) instead of relying on a shared start-up seed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(value)
    on_dev = dist.get_backend(group) == 'nccl'
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if on_dev else 'cpu')
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.cpu()[0])


_NATIVE_COMMS = {}
CANARY_LOG = []          # one dict per native_comm() attempt of this process (bench.py reports the last one)


def forget_native_comm(comm):
    """Drop every cached reference to the communicator at address `comm` (the library aborted it inside a call, MFG_ECOMM):
    later native_comm() calls for the same group then answer None -- the exchange stays in torch.distributed -- instead of
    handing out a dead handle."""
    for key, val in list(_NATIVE_COMMS.items()):
        if val == comm:
            _NATIVE_COMMS[key] = None


def _canary(lib, comm, device, world, rank, timeout):
    """Prove the communicator before the class relies on it: in a helper thread, on a side stream, (1) one all-reduce of a
    known pattern (rank + 1 -> world (world + 1) / 2) and (2) a tiny mfg_train_rollouts_dist run (4 trajectories per rank,
    d = 21, T = 2, three episodes = three collectives inside the native loop), checked for finite parameters and for the
    all-reduced sample count 4 * world * T.  Returns (ok, thread_still_running, detail, the thread)."""
    import threading
    from . import ops
    res = {}

    def body():
        try:
            with torch.cuda.device(device):
                side = torch.cuda.Stream(device)
                with torch.cuda.stream(side):
                    buf = torch.full((4,), float(rank + 1), dtype=torch.float64, device=device)
                    rc = lib.mfg_dist_all_reduce(comm, buf.data_ptr(), 4, side.cuda_stream)
                    side.synchronize()
                    want = world * (world + 1) / 2.0
                    if rc != 0 or not bool((buf == want).all()):
                        res['detail'] = 'all-reduce: rc %d, got %r, want %r' % (rc, buf.tolist(), want)
                        return
                    d, T, Bl, eps = 21, 2, 4, 3
                    F = ops.num_features(d)
                    mat = torch.full((2, d), 1.0 / d, dtype=torch.float32, device=device)
                    theta = torch.tensor([8.86349], dtype=torch.float64, device=device)
                    w = torch.full((F,), 0.5, dtype=torch.float64, device=device)
                    ta, wa = torch.empty_like(theta), torch.empty_like(w)
                    G = torch.zeros(F + 3, dtype=torch.float64, device=device)
                    ws = ops.workspace(Bl * T, d, device)
                    bufs = {'pi_traj': torch.empty(Bl, T + 1, d, dtype=torch.float32, device=device),
                            'pi_last': torch.empty(Bl, d, dtype=torch.float32, device=device),
                            'reward': torch.empty(Bl, T, dtype=torch.float32, device=device),
                            'delta': torch.empty(Bl, T, dtype=torch.float64, device=device),
                            'g': torch.empty(Bl, T, dtype=torch.float64, device=device)}
                    ops.train_rollouts_dist(comm, mat, T, eps, 0, False, theta, w, ta, wa, 0.16, 12000.0, 1.0, G, ws, bufs,
                                            0.1, 0.001, seed=11, first_step=0, traj_offset=rank * Bl)
                    side.synchronize()
                    count = float(G[F + 2])
                    fin = bool(torch.isfinite(theta).all() and torch.isfinite(w).all())
                    if count != float(Bl * world * T) or not fin:
                        res['detail'] = 'native loop: count %r (want %r), finite %r' % (count, Bl * world * T, fin)
                        return
                    res['theta'] = float(theta[0])
            res['ok'] = True
        except Exception as exc:                       # noqa: BLE001 -- any failure means "do not use the native loop"
            res['detail'] = repr(exc)

    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(timeout)
    inject = os.environ.get('MFG_NATIVE_RCCL_CANARY_FAIL', '')     # test hook: 'all' or a rank number fails its (healthy) canary
    if inject and (inject == 'all' or inject == str(rank)):
        return False, th.is_alive(), 'injected failure (MFG_NATIVE_RCCL_CANARY_FAIL=%s)' % inject, th
    return bool(res.get('ok')) and not th.is_alive(), th.is_alive(), res.get('detail', 'timed out' if th.is_alive() else ''), th


def native_comm(group=None, device=None, allow_single=False):
    """Address of an RCCL communicator owned by the HIP library for `group` (None where it cannot be had): the multi-GPU
    episode loop then issues its one all-reduce per update itself (mfg_train_rollouts_dist) instead of returning to Python
    and torch.distributed for every episode.  Rank 0 obtains an ncclUniqueId from the library, it is broadcast through the
    existing process group, every rank initialises the communicator on `device`.  Only for jobs whose process group runs on
    RCCL ('nccl' backend: one GPU per rank); created once per group and kept for the life of the process.

    SELF-ENABLING behind a canary (round 6; it was opt-in through MFG_NATIVE_RCCL=1 before): no multi-GPU node was available
    to any build so far, so the loop proves itself on the job's own communicator before the class uses it, and every rank
    falls back to torch.distributed TOGETHER if any rank's proof fails.  MFG_NATIVE_RCCL=0 switches it off (all ranks must
    agree on the setting; a rank that has it off still takes part in the first agreement, so a mixed setting ends in a common
    fallback instead of a hang).

    Protocol -- nothing rank-local may fail between "everybody agreed to try" and a blocking collective, so every step ends
    in an agreement (all-reduce MIN through the process group, which has the job's timeout):
      1. "my library resolved RCCL and I want the native loop" (rank 0: and hands out an id)       -> agree
      2. ncclCommInitRank in a helper thread, bounded by MFG_DIST_INIT_TIMEOUT (default 60 s)       -> agree
      3. the canary (_canary: one patterned all-reduce + a tiny mfg_train_rollouts_dist), helper thread, same bound -> agree
    After a failed agreement a rank whose own communicator is fine ABORTS it (ncclCommAbort: no hand-shake with peers that may
    never arrive); a helper thread that finishes after its time-out aborts what it made itself (nobody else knows the handle).
    A rank whose canary is still stuck after the abort exits non-zero (os._exit(70)): its stream is wedged and torch's watchdog
    would only find out at the next collective.  Nothing is ever re-executed."""
    import ctypes as C
    import threading
    from . import _lib as L
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size(group)
    if dist.get_backend(group) != 'nccl' or (world == 1 and not allow_single):
        return None
    key = (id(group), world)
    if key in _NATIVE_COMMS:
        return _NATIVE_COMMS[key]
    wanted = os.environ.get('MFG_NATIVE_RCCL', '1') != '0'
    device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    lib = L.lib()
    rank = dist.get_rank(group)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    timeout = float(os.environ.get('MFG_DIST_INIT_TIMEOUT', '60'))
    log = {'world': world, 'rank': rank, 'wanted': wanted, 'stage': 'resolve', 'native': False, 'detail': ''}
    CANARY_LOG.append(log)

    def agree(flag):
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.cpu()[0]) == 1

    def abort(handle):
        if handle:
            try:
                lib.mfg_dist_abort(handle)
            except Exception:                          # noqa: BLE001
                pass

    comm = None
    with torch.cuda.device(device):                                     # ncclCommInitRank binds to the CURRENT device
        buf = (C.c_char * 128)()
        # 1. every rank wants the native loop and its library can resolve RCCL (rank 0: and hand out an id) -- a rank that
        #    cannot says so BEFORE anybody blocks
        mine = wanted and (lib.mfg_dist_unique_id(buf) == 0 if rank == 0 else lib.mfg_dist_available() == 1)
        if agree(mine):
            log['stage'] = 'init'
            t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone().to(device)
            dist.broadcast(t, src=src, group=group)
            idb = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
            out = C.c_void_p()
            result = {}
            lock = threading.Lock()

            def init():
                with torch.cuda.device(device):
                    rc = lib.mfg_dist_init(idb, world, rank, C.byref(out))
                with lock:
                    result['rc'] = rc
                    late = result.get('cancelled', False)
                if late and rc == 0 and out.value:     # finished after the time-out: nobody owns this communicator
                    abort(out.value)
            torch.cuda.synchronize(device)
            th = threading.Thread(target=init, daemon=True)
            th.start()
            th.join(timeout)
            with lock:
                done = 'rc' in result
                if not done:
                    result['cancelled'] = True
            if done and result.get('rc') == 0 and out.value:
                comm = out.value
            # 2. a rank that failed or timed out must not leave the others using a communicator it is not part of
            if not agree(comm is not None):
                log['detail'] = 'ncclCommInitRank failed or timed out on some rank'
                abort(comm)
                comm = None
            else:
                # 3. the canary
                log['stage'] = 'canary'
                ok, stuck, detail, cth = _canary(lib, comm, device, world, rank, timeout)
                log['detail'] = detail
                if not agree(ok):
                    abort(comm)
                    comm = None
                    if stuck:
                        cth.join(10.0)
                        if cth.is_alive():
                            import sys
                            sys.stderr.write('discrete_mean_field_game_amd: rank %d: the native RCCL canary did not return after '
                                             'ncclCommAbort; exiting\n' % rank)
                            sys.stderr.flush()
                            os._exit(70)
                else:
                    log['native'] = True
                    log['stage'] = 'ok'
    _NATIVE_COMMS[key] = comm
    return comm


def lr_scales(episode: int, constant) -> tuple:
    """(critic, actor) learning-rate multipliers of the reference schedule in `episode`
    (mfg_ac2.py:511-522: 1/(episode+1) and 1/((episode+1) ln ln(episode+20)); 1, 1 if constant)."""
    if constant:
        return 1.0, 1.0
    return 1.0 / (episode + 1), 1.0 / ((episode + 1) * math.log(math.log(episode + 20)))
