"""CPU: pins the oracle's rescale restatement against big integers, as ring/ring_test.go:242-331 does
(testDivFloorByLastModulusMany / testDivRoundByLastModulusMany: random x < Q/10, nbRescales = level)."""
import numpy as np
import pytest

from conftest import QI60
from test_oracle_bext import prod, rns


def div_round(a, b):
    return (2 * a + b) // (2 * b)


@pytest.mark.parametrize("round_", [0, 1])
@pytest.mark.parametrize("L,nb", [(4, 3), (6, 1), (6, 5), (2, 1)])
def test_div_by_last_modulus_many_vs_bigint(oracle, L, nb, round_):
    Q = QI60[:L]
    rng = np.random.default_rng(L * 10 + nb + round_)
    n = 64
    big = prod(Q)
    vals = [int.from_bytes(rng.bytes(big.bit_length() // 8 + 2), "little") % big // 10 for _ in range(n)]
    want = list(vals)
    for j in range(nb):
        m = Q[L - 1 - j]
        want = [div_round(v, m) if round_ else v // m for v in want]
    out = oracle.div_by_last_modulus_many(rns(vals, Q), Q, nb, round_)
    assert out.shape[0] == L - nb
    for i in range(L - nb):
        assert [int(x) for x in out[i]] == [w % Q[i] for w in want]
