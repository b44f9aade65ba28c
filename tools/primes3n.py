"""Moduli for the 3N-ring measurement scripts, without the test oracle: the stepping rule of Find3NRNSPrimes (ring/primes_3n.go:11-43,
candidates 1 mod 3N upward from 2^bits) with a deterministic Miller-Rabin test for 64-bit integers."""


def is_prime(n):
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2; s += 1
    for a in small:                                   # these bases decide every n < 3.3 * 10^24
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def moduli_3n(N, count, bits=60):
    """the first `count` primes q = 1 mod 3N above 2^bits"""
    step = 3 * N
    c = ((1 << bits) // step + 1) * step + 1
    out = []
    while len(out) < count:
        if is_prime(c):
            out.append(c)
        c += step
    return out
