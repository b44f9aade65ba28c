#!/usr/bin/env python3
"""Generates tests/golden/ntt3n_intdft.json by IMPORTING the reference's Python notes references/integer_dft.py
(build container only; the reference tree does not exist on the GPU box).

For each N: the prime p and 3N-th root w that IntegerDFT picks, a seeded input, factorized_dft(input) in the
reference's TREE order, the last tree level (exponent of w evaluated at each slot), and factorized_idft round trip.
The fixture holds data only."""
import importlib.util
import json
import os
import random
import sys

sys.dont_write_bytecode = True
REF = "/root/reference/references/integer_dft.py"
spec = importlib.util.spec_from_file_location("ref_integer_dft", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

out = {"source": "references/integer_dft.py (IntegerDFT.factorized_dft / factorized_idft, tree)", "vectors": []}
rnd = random.Random(20261004)
for N in (6, 12, 18, 24, 36, 48, 96, 192):
    d = mod.IntegerDFT(N, min_bits=16)
    x = [rnd.randrange(d.p) for _ in range(N)]
    y = [int(v) for v in d.factorized_dft(x)]
    back = [int(v) for v in d.factorized_idft(y)]
    assert back == x
    out["vectors"].append({"N": N, "p": int(d.p), "w": int(d.w), "input": x, "dft_tree_order": y,
                           "tree_last": [int(v) for v in d.tree[d.level]]})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ntt3n_intdft.json")
json.dump(out, open(path, "w"))
print("wrote", path, [(v["N"], v["p"]) for v in out["vectors"]])
