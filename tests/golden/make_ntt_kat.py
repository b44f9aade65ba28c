#!/usr/bin/env python3
"""Extracts the known-answer vectors of the reference's ring/ntt_test.go (testVector, lines 10-89) into ntt_kat.json.

Run in the build container only (reads /root/reference as TEXT; nothing is imported or executed from it):
    python tests/golden/make_ntt_kat.py
The output holds data only: N, the moduli, the input limbs (`poly`) and the expected forward NTT (`polyNTT`)."""
import json, os, re, sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/ring/ntt_test.go"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ntt_kat.json")

text = open(SRC).read()
start = text.index("var testVector")
end = text.index("func TestNTT")
body = text[start:end]

def limbs(block):
    rows = re.findall(r"\{([0-9,\s]+)\}", block)
    return [[int(x) for x in r.replace("\n", " ").split(",") if x.strip()] for r in rows]

vectors = []
# each entry: { N, []uint64{q...}, Poly{Coeffs: [][]uint64{ {..}, {..} }}, Poly{Coeffs: ...} }
for m in re.finditer(r"\{\s*(\d+),\s*\[\]uint64\{([^}]*)\},(.*?)\n\t\},", body, re.S):
    N = int(m.group(1))
    qs = [int(x) for x in m.group(2).split(",") if x.strip()]
    rest = m.group(3)
    parts = rest.split("Poly{")
    assert len(parts) == 3, len(parts)
    poly, polyntt = limbs(parts[1]), limbs(parts[2])
    assert len(poly) == len(qs) == len(polyntt)
    assert all(len(l) == N for l in poly + polyntt), (N, [len(l) for l in poly + polyntt])
    vectors.append({"N": N, "Qis": qs, "poly": poly, "polyNTT": polyntt})

assert [v["N"] for v in vectors] == [16, 32, 64, 128, 256, 512], [v["N"] for v in vectors]
json.dump({"source": "ring/ntt_test.go:10-89 (testVector)", "vectors": vectors}, open(OUT, "w"))
print("wrote", OUT, "with", len(vectors), "vectors")
