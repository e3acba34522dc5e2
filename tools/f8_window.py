#!/usr/bin/env python3
"""Diagnostic: how far below the largest product of one 128-deep fp8 MFMA a small product survives."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f8_precision import run  # noqa: E402
from composable_kernel_amd import capi  # noqa: E402


def main():
    M = N = 256
    K = 256
    for pos in (1, 40, 130):        # same 32-block, other 32-block of the same MFMA, next MFMA
        line = []
        for e_small in range(1, 16):
            a = np.zeros((M, K), np.uint8)
            q = np.zeros((N, K), np.int8)
            a[:, 0] = 0x78            # 256
            a[:, pos] = e_small << 3  # 2^(e_small - 7)
            q[:, 0] = 7
            q[:, pos] = 1
            got, _ = run(a, q, M, N, K, capi.F8_E4M3FN)
            exact = 256.0 * 7 + 2.0 ** (e_small - 7)
            line.append("ok" if got[0, 0] == np.float32(exact) else f"{got[0,0]-1792:.6g}!")
        print(f"small term at k={pos:3d}: exps 1..15 ->", " ".join(line))


if __name__ == "__main__":
    main()
