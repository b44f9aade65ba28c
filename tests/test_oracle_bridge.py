"""CPU: the oracle's restatement of ring/conjugate_invariant.go on a case worked by hand (n = 8, q = 97) and the closed form of the
reference's in-place Pad loop that the device kernel and the large-size GPU test use."""
import numpy as np
import pytest

from oracle import compose


def test_bridge_restatements_by_hand():
    c8 = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.uint64)
    s16 = np.array([(90 + i) % 97 for i in range(16)], dtype=np.uint64)
    assert compose.unfold_ci_to_standard(c8).tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 8, 7, 6, 5, 4, 3, 2, 1]
    assert compose.fold_standard_to_ci(s16, [15, 14, 13, 12, 11, 10, 9, 8], 97).tolist() == [1] * 8
    assert compose.pad_default_to_ci(c8, True, 97, s16).tolist() == [1, 2, 3, 4, 4, 3, 2, 1] + s16[8:].tolist()
    assert compose.pad_default_to_ci(c8, False, 97, s16).tolist() == [0, 2, 3, 4, 92, 93, 94, 95] + s16[8:].tolist()


@pytest.mark.parametrize("n", [2, 4, 16, 256])
def test_pad_closed_form_equals_the_in_place_loop(n):
    rng = np.random.default_rng(n)
    q = 0x1fffffffffe00001
    x = rng.integers(0, q, size=n, dtype=np.uint64)
    x[rng.integers(0, n)] = 0
    before = rng.integers(0, q, size=2 * n, dtype=np.uint64)
    h = n // 2
    ntt = compose.pad_default_to_ci(x, True, q, before)
    assert np.array_equal(ntt[:h], x[:h]) and np.array_equal(ntt[h:n], x[:h][::-1]) and np.array_equal(ntt[n:], before[n:])
    cf = compose.pad_default_to_ci(x, False, q, before)
    exp = np.empty(n, dtype=np.uint64)
    exp[0] = 0
    exp[1:h] = x[1:h]
    exp[h] = np.uint64(q) - x[h]
    exp[h + 1:] = (np.uint64(q) - x[1:h])[::-1]
    assert np.array_equal(cf[:n], exp) and np.array_equal(cf[n:], before[n:])
