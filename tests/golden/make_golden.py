"""Generates tests/golden/lsb_golden.npz (committed fixture).

Inputs: the key stream lsb/cub/test/test_device_radix_sort.cu feeds its tests
(RANDOM mode = RandomBits over MT19937 seeded {0x123,0x234,0x345,0x456},
lsb/cub/test/test_util.h:96-97,408-458).  The MT19937 words are produced by
the REFERENCE's own lsb/cub/test/mersenne.h, compiled from where it lies into
oracle/_ref/libref_mersenne.so (oracle/Makefile `ref`); RandomBits' AND
reduction is applied here in numpy.  Expected outputs: numpy stable argsort on
the masked key -- the InitializeSolution rule (test_device_radix_sort.cu
:634-693): an independent implementation, NOT oracle.c and NOT the GPU path.

Run from the repo root in the build container:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "libref_mersenne.so")

SIZES = [0, 1, 2, 255, 256, 257, 6911, 6912, 6913, 8191, 8192, 8193, (1 << 14) + 3]
FULL = ((0, 32), (1, 31), (15, 17))


def ref_stream(nwords):
    R = C.CDLL(REF)
    R.ref_mt_genrand_int32.restype = C.c_uint32
    init = (C.c_uint32 * 4)(0x123, 0x234, 0x345, 0x456)
    R.ref_mt_init_by_array(init, 4)
    return np.array([R.ref_mt_genrand_int32() for _ in range(nwords)], dtype=np.uint32)


def random_bits(stream, n, entropy_reduction):
    """RandomBits<unsigned int>: AND of (entropy_reduction+1) consecutive words per key."""
    w = stream[: n * (entropy_reduction + 1)].reshape(n, entropy_reduction + 1)
    return np.bitwise_and.reduce(w, axis=1).astype(np.uint32)


def expected(keys, vals, begin_bit, end_bit, descending):
    nb = end_bit - begin_bit
    masked = keys if nb >= 32 else (keys & np.uint32(((1 << nb) - 1) << begin_bit))
    sort_key = (~masked).astype(np.uint32) if descending else masked
    order = np.argsort(sort_key, kind="stable")
    return keys[order], vals[order]


def main():
    nmax = max(SIZES)
    stream = ref_stream(nmax * 4)
    out = {"sizes": np.array(SIZES, dtype=np.int64), "mt_first8": stream[:8]}
    cases = []
    for er in (0, 3):
        base = random_bits(stream, nmax, er)
        out[f"keys_er{er}"] = base
        for n in SIZES:
            keys = base[:n]
            vals = np.arange(n, dtype=np.uint32)
            if n <= 257:
                combos = [(bb, eb, d) for (bb, eb) in FULL for d in (0, 1)]
            elif er == 0:
                combos = [(0, 32, 0), (0, 32, 1), (1, 31, 0), (15, 17, 1)]
            else:
                combos = [(0, 32, 0)]
            for (bb, eb, desc) in combos:
                k, v = expected(keys, vals, bb, eb, desc)
                tag = f"er{er}_n{n}_b{bb}_{eb}_d{desc}"
                out["v_" + tag] = v          # expected permutation; keys = input[v]
                cases.append(tag)
    out["cases"] = np.array(cases)
    path = os.path.join(ROOT, "tests", "golden", "lsb_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
