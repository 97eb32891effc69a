"""CPU: the NumPy restatement of the engine's counter RNG (oracle/philox.py)."""
import numpy as np

from oracle.philox import philox4x32_10, u53, PhiloxStream


def test_random123_known_answers():
    # kat_vectors of Random123 (philox4x32, 10 rounds)
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(x) for x in philox4x32_10(*ctr, *key))
        assert got == want


def test_u53_range_and_exactness():
    a = np.array([0, 0xffffffff, 0x12345678], dtype=np.uint32)
    b = np.array([0, 0xffffffff, 0x9abcdef0], dtype=np.uint32)
    u = u53(a, b)
    assert u[0] == 2.0 ** -53 and u[1] == 1.0 and 0 < u[2] < 1


def test_streams_are_keyed_by_particle_id_not_position():
    whole = PhiloxStream(99, np.arange(100))
    part = PhiloxStream(99, np.arange(40, 60))
    assert np.array_equal(whole.normals(7, 3)[:, 40:60], part.normals(7, 3))
    assert np.array_equal(whole.unit_exponentials(5)[:, 40:60], part.unit_exponentials(5))
    assert not np.array_equal(whole.normals(7, 3), whole.normals(7, 4))


def test_moments():
    s = PhiloxStream(1, np.arange(20000))
    z = s.normals(8, 1)
    e = s.unit_exponentials(1)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(e.mean() - 1) < 0.02 and e.min() >= 0
