"""NumPy restatement of the device RNG: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).

TEST INFRASTRUCTURE.  The reference draws its gamma variates from NumPy's legacy MT19937 stream
(mfg_ac2.py:242), which is inherently serial; the batched framework replaces it by a counter-based
generator (SURVEY.md section 7 "RNG parity").  This file pins the integer part of that design bit for bit:
the known-answer vectors of the Random123 distribution are checked in tests/test_philox_ref.py and the
HIP kernel's raw output is compared against philox4x32_10() in the -m gpu tests.

Counter layout used by the kernels (csrc/mfg_device.h::philox_elem):
    c0 = element i*d + j, c1 = env step, c2 = low 32 bits of the global trajectory id,
    c3 = (high 16 bits of the trajectory id) | (draw block << 16);  key = (seed low, seed high).
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy arrays of uint32 counters.  Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64); c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64); c3 = np.asarray(c3, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def philox_elem(seed, elem, step, traj, block):
    """Counter layout of the samplers (see module docstring)."""
    traj = np.asarray(traj, dtype=np.uint64)
    c2 = traj & MASK
    c3 = ((traj >> np.uint64(32)) & np.uint64(0xFFFF)) | (np.asarray(block, dtype=np.uint64) << np.uint64(16))
    return philox4x32_10(elem, step, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u01(r):
    """(0,1] float32 from the top 24 bits (csrc/mfg_device.h::u01)."""
    return ((r >> np.uint32(8)).astype(np.float64) * 2.0 ** -24 + 2.0 ** -25).astype(np.float32)   # one fma rounding


START_DRAW_ELEM = 0xFFFFFFFF


def start_indices(seed, step, traj, num_start):
    """Start-state rows of batched runs (csrc/mfg_device.h::start_draw_row, include/mfg_hip.h mfg_draw_start): the
    reference's per-episode `idx_row = np.random.randint(num_start_samples)` (mfg_ac2.py:466, ac_irl.py:655) drawn from the
    counter-based generator instead -- counter (0xFFFFFFFF, step of the episode's first env step, global trajectory id,
    block 0), idx = floor(x0 * num_start / 2^32).  Integer bookkeeping: bit exact against the kernels."""
    x0 = philox_elem(int(seed), START_DRAW_ELEM, int(step), np.asarray(traj, dtype=np.uint64), 0)[0]
    return ((x0.astype(np.uint64) * np.uint64(int(num_start))) >> np.uint64(32)).astype(np.int64)
