#!/usr/bin/env python3
"""Generates tests/golden/ring3n_product.json by IMPORTING the reference's Python notes references/integer.py (build container only):
cyclotomic_polynomial_multiply_ntt(a, b) = a * b in Z_p[X]/(X^N - X^(N/2) + 1) through the reference's own factorized NTT, cross-checked here
against its naive_cyclotomic_multiply at N = 48.

What it pins: schemes/matrix_ckks/evaluator.go:114-192 (Evaluator.Mul) has no test vector in the reference.  Its Go code multiplies
NTT-domain operands with MulCoeffsMontgomery and never MForm's them, so every output coefficient is the ring product times 2^-64 mod p; with
degree-1 inputs (a0, a1), (b0, b1) the three outputs are a0 b0, a0 b1 + a1 b0, a1 b1.  The fixture stores those three ring products (S0, S1, S2)
as the reference's Python computes them -- SHA-256 + 256 spot values per size; inputs are rules, not stored:
    a0[i] = (5 i^2 + i + 2) mod p, a1[i] = (i^3 + 7) mod p, b0[i] = (3 i^2 + 11 i + 1) mod p, b1[i] = (i^2 + 13 i + 5) mod p."""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/references")
spec = importlib.util.spec_from_file_location("ref_integer", "/root/reference/references/integer.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
from integer_dft import IntegerDFT      # noqa: E402  (the reference's module, found through the path above)


def inputs(N, p):
    i = np.arange(N, dtype=object)
    return ([int(v) for v in (5 * i * i + i + 2) % p], [int(v) for v in (i * i * i + 7) % p],
            [int(v) for v in (3 * i * i + 11 * i + 1) % p], [int(v) for v in (i * i + 13 * i + 5) % p])


out = {"source": "references/integer.py cyclotomic_polynomial_multiply_ntt (IntegerDFT factorized NTT), min_bits = 16", "vectors": []}
for N in (48, 768, 3 << 13, 3 << 16):
    p = int(IntegerDFT(N, min_bits=16).p)
    a0, a1, b0, b1 = inputs(N, p)
    mul = lambda x, y: [int(v) for v in mod.cyclotomic_polynomial_multiply_ntt(x, y, min_bits=16)]
    S0, P01, P10, S2 = mul(a0, b0), mul(a0, b1), mul(a1, b0), mul(a1, b1)
    S1 = [(u + v) % p for u, v in zip(P01, P10)]
    if N == 48:
        assert S0 == [int(v) % p for v in mod.naive_cyclotomic_multiply(a0, b0, p)]
    spots = [int(v) for v in np.linspace(0, N - 1, 256 if N > 256 else N).astype(np.int64)]
    e = {"N": N, "p": p, "spots": spots}
    for name, S in (("S0", S0), ("S1", S1), ("S2", S2)):
        arr = np.array(S, dtype=np.uint64)
        e[name + "_sha256_u64le"] = hashlib.sha256(arr.astype("<u8").tobytes()).hexdigest()
        e[name + "_at_spots"] = [int(arr[s]) for s in spots]
    out["vectors"].append(e)
    print(N, p, e["S0_sha256_u64le"][:16])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ring3n_product.json")
json.dump(out, open(path, "w"), indent=0)
print("wrote", path, os.path.getsize(path))
