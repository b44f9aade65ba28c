#!/usr/bin/env python3
"""Generates tests/golden/ntt3n_intdft_large.json (+ .npz) by IMPORTING the reference's Python notes references/integer_dft.py (build
container only; the reference tree does not exist on the GPU box) at the sizes the BASELINE configs use: N = 3*2^13 (config 2) in full,
N = 3*2^14, 3*2^15, 3*2^16 (config 4's ring) and N = 9*2^10 (two radix-3 layers) as SHA-256 + 256 spot values each.

Per size: the prime p and the 3N-th root w IntegerDFT picks (min_bits = 16: p has 18-22 bits), the input x[i] = (7 i^2 + 3 i + 1) mod p
(a rule, not stored), y = factorized_dft(x) in the reference's TREE order, and `ascending` = y re-ordered by ascending exponent of w --
slot s of the tree order evaluates at w^tree[level][s]; sorting those exponents gives the order of the Go transformer
(ring/ntt_3n.go:82-109, 235-243).  The re-ordering is an argsort of the reference's own tree: the fixture holds data only."""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/references/integer_dft.py"
spec = importlib.util.spec_from_file_location("ref_integer_dft", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

here = os.path.dirname(os.path.abspath(__file__))
meta = {"source": "references/integer_dft.py (IntegerDFT.factorized_dft / factorized_idft, tree), input rule x[i] = (7 i^2 + 3 i + 1) mod p",
        "vectors": []}
arrays = {}
for N, full in ((3 << 13, True), (9 << 10, False), (3 << 14, False), (3 << 15, False), (3 << 16, False)):
    d = mod.IntegerDFT(N, min_bits=16)
    p = int(d.p)
    x = [(7 * i * i + 3 * i + 1) % p for i in range(N)]
    y = np.array([int(v) for v in d.factorized_dft(x)], dtype=np.uint64)
    assert [int(v) for v in d.factorized_idft([int(v) for v in y])] == x
    exps = np.array([int(v) for v in d.tree[d.level]], dtype=np.int64)
    asc = y[np.argsort(exps, kind="stable")]
    spots = [int(v) for v in np.linspace(0, N - 1, 256).astype(np.int64)]
    entry = {"N": N, "p": p, "w": int(d.w), "sha256_ascending_u64le": hashlib.sha256(asc.astype("<u8").tobytes()).hexdigest(),
             "sha256_tree_order_u64le": hashlib.sha256(y.astype("<u8").tobytes()).hexdigest(),
             "spots": spots, "ascending_at_spots": [int(asc[s]) for s in spots], "full": bool(full)}
    if full:
        arrays["ascending_%d" % N] = asc.astype(np.uint32)
        arrays["tree_order_%d" % N] = y.astype(np.uint32)
        arrays["tree_last_%d" % N] = exps.astype(np.uint32)
    meta["vectors"].append(entry)
    print(N, p, int(d.w), entry["sha256_ascending_u64le"][:16])
json.dump(meta, open(os.path.join(here, "ntt3n_intdft_large.json"), "w"), indent=0)
np.savez_compressed(os.path.join(here, "ntt3n_intdft_large.npz"), **arrays)
print("wrote", os.path.join(here, "ntt3n_intdft_large.json"), os.path.getsize(os.path.join(here, "ntt3n_intdft_large.json")),
      os.path.getsize(os.path.join(here, "ntt3n_intdft_large.npz")))
