"""Numerics of the three-piece bf16 split (x = x1 + x2 + x3, six bf16 x bf16 products with fp32
accumulation) for the two GEMM-shaped loss terms: G = F F^T / n (Gram) and S = D F (SYMM).
Emulated in numpy: bf16 pieces by round-to-nearest-even of the residuals, every product exact
(8 x 8 significand bits fit fp32), sums carried in float64 and rounded to float32 once per
K-block of `kb` (an optimistic model of the MFMA's fp32 accumulator) and, pessimistically, in
plain float32 cumulative order.  Reference: float64.  Printed: max |err| / max |ref|.

    python tools/bf16x3_numerics.py
"""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def split3(x):
    x1 = bf16_rne(x)
    r = (x - x1).astype(np.float32)
    x2 = bf16_rne(r)
    x3 = bf16_rne((r - x2).astype(np.float32))
    return x1, x2, x3


PAIRS = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]      # terms down to 2^-16 relative


def matmul_split(a, b, kb=512):
    """sum over PAIRS of a_i @ b_j with fp32 rounding after every kb of K (per pair, then added
    smallest first)."""
    ap, bp = split3(a), split3(b)
    k = a.shape[1]
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for i, j in reversed(PAIRS):
        part = np.zeros_like(acc)
        for k0 in range(0, k, kb):
            part = (part.astype(np.float64) + ap[i][:, k0:k0 + kb].astype(np.float64)
                    @ bp[j][k0:k0 + kb].astype(np.float64)).astype(np.float32)
        acc = (acc + part).astype(np.float32)
    return acc


def matmul_f32_blocks(a, b, kb=512):
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k0 in range(0, a.shape[1], kb):
        acc = (acc.astype(np.float64) + a[:, k0:k0 + kb].astype(np.float64)
               @ b[k0:k0 + kb].astype(np.float64)).astype(np.float32)
    return acc


def rel(x, ref):
    return float(np.abs(x.astype(np.float64) - ref).max() / np.abs(ref).max())


rng = np.random.RandomState(0)
for c, hw in ((64, 65536), (128, 16384), (512, 4096)):
    f = np.maximum(rng.standard_normal((c, hw)) * 30 + 5, 0).astype(np.float32)   # post-ReLU, biased
    ref = f.astype(np.float64) @ f.astype(np.float64).T
    print('Gram C=%d HW=%d:  split %.2e   fp32 blocks %.2e   numpy f32 %.2e   (two pieces only: %.2e)'
          % (c, hw, rel(matmul_split(f, f.T), ref), rel(matmul_f32_blocks(f, f.T), ref),
             rel(f @ f.T, ref),
             rel(sum(np.float64(a) @ np.float64(b).T for a in split3(f)[:2] for b in split3(f)[:2]), ref)))
    d = rng.standard_normal((c, c)).astype(np.float32)
    d = np.tril(d) + np.tril(d, -1).T
    ref = d.astype(np.float64) @ f.astype(np.float64)
    print('SYMM C=%d HW=%d:  split %.2e   fp32 blocks %.2e   numpy f32 %.2e'
          % (c, hw, rel(matmul_split(d, f, kb=64), ref), rel(matmul_f32_blocks(d, f, kb=64), ref),
             rel(d @ f, ref)))
