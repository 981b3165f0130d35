"""Known-answer tests of the Philox4x32-10 restatement (Random123 kat_vectors)."""
import numpy as np

from oracle.philox_ref import philox4x32_10, u01


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
