"""Numpy emulation of the wave-level MFMA semantics used by the HIP kernels (test helper).

Operand images are [64 lanes][J] arrays, lane = hi*32 + n.  D = sum_{hi,j} A[hi*32+i][j]*B[hi*32+n][j];
D registers: lane (n,hi), register r holds D[row = (r&3)+8*(r>>2)+4*hi][col = n]
(cdna_hip_programming.md section 3, C/D map of the 32x32 shapes)."""
import numpy as np


def drow(r, hi):
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def mfma(A, B, C):
    """A, B: (64, J) operand images; C: (64,16) register image -> (64,16) float32 (fp32 accumulate)."""
    A = A.astype(np.float64).reshape(2, 32, -1)
    B = B.astype(np.float64).reshape(2, 32, -1)
    D = np.einsum("hij,hnj->in", A, B)            # [row i][col n]
    out = C.astype(np.float64).copy()
    for hi in range(2):
        for r in range(16):
            out[hi * 32:(hi + 1) * 32, r] += D[drow(r, hi), :]
    return out.astype(np.float32)
