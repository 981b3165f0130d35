"""Known-answer tests of the Philox4x32-10 restatement (Random123 kat_vectors)."""
import numpy as np

from oracle.philox_ref import philox4x32_10, start_indices, u01


def test_random123_known_answers():
    # Random123 examples/kat_vectors: philox4x32 10 rounds
    r = philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_u01_never_zero():
    # (0, 1]: log(u) is always finite; the +0.5 is absorbed by fp32 rounding above 2^23
    r = np.array([0, 0xFFFFFFFF, 0x80000000, 0x100], dtype=np.uint32)
    u = u01(r)
    assert u.dtype == np.float32 and u.min() > 0 and u.max() <= 1
    assert u[0] == np.float32(2.0 ** -25) and u[2] == np.float32(0.5) and u[3] == np.float32(1.5 * 2.0 ** -24)
    assert u[1] == np.float32(1.0)


def test_start_indices_are_the_documented_philox_words():
    """include/mfg_hip.h mfg_draw_start: counter (0xFFFFFFFF, step, trajectory id low, trajectory id bits 32..47), key = seed,
    idx = floor(x0 * num_start / 2^32) -- spelled out here with the raw block function (integer bookkeeping)."""
    seed, step = 0x0123456789ABCDEF, 4500
    traj = np.array([0, 1, 2, 65535, 2 ** 32 - 1, 2 ** 32, 5 * 2 ** 32 + 7, 2 ** 47 + 3], dtype=np.uint64)
    for n in (1, 2, 3, 64, 1000, 2 ** 31 - 1):
        got = start_indices(seed, step, traj, n)
        x0 = philox4x32_10(0xFFFFFFFF, step, traj & np.uint64(0xFFFFFFFF), (traj >> np.uint64(32)) & np.uint64(0xFFFF),
                           seed & 0xFFFFFFFF, seed >> 32)[0]
        want = np.array([(int(x) * n) >> 32 for x in x0])
        assert got.dtype == np.int64 and np.array_equal(got, want)
        assert got.min() >= 0 and got.max() < n
    # the Random123 known answer with counter word 0 = 0xFFFFFFFF: all-ones counter and key give x0 = 0x408f276d; the start
    # draw's counter differs from it only in c3 (block 0 -> upper 16 bits zero), so it must NOT reproduce that word
    assert int(philox4x32_10(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)[0]) == 0x408f276d


def test_start_indices_are_uniform_and_independent_of_the_split():
    n, B = 64, 1 << 16
    idx = start_indices(7, 15, np.arange(B), n)
    counts = np.bincount(idx, minlength=n)
    chi2 = float(((counts - B / n) ** 2 / (B / n)).sum())
    assert chi2 < 120.0                                              # 63 degrees of freedom: mean 63, sd 11.2
    # a rank's shard draws exactly its slice of the global vector; another episode (step) is a different vector
    assert np.array_equal(start_indices(7, 15, np.arange(1000, 3000), n), idx[1000:3000])
    assert not np.array_equal(start_indices(7, 30, np.arange(B), n), idx)
    assert not np.array_equal(start_indices(8, 15, np.arange(B), n), idx)
