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


def main_gemm():
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


# ------------------------------------------------------------------------------------------------
# The same split inside a 3x3 convolution (csrc/conv_bf3.hip, off by default): 1-D Winograd F(2,3)
# along x, V = B^T d and U = G g formed in float32, both split into three bf16 pieces, the six
# products of a 16-channel step added to a float32 accumulator one MFMA at a time (each MFMA's
# sixteen products exactly, one rounding when they join the accumulator -- the order the kernel
# uses: x2y2, x1y3, x3y1, x1y2, x2y1, x1y1; kernel rows ky = 0, 1, 2 within a chunk).  `hilo`: the
# five small products in an accumulator of their own, added at the end.  Reference: float64 direct
# convolution.  Next to it the float32 2-D Winograd F(2x2, 3x3) the engine ships (accumulation
# modelled the same way, two channels per MFMA).
def conv_bf3_emulated(x, w, hilo):
    cin, h, wd = x.shape
    cout = w.shape[0]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1 + wd % 2)))
    t = (wd + 1) // 2
    d = np.stack([xp[:, :, i:i + 2 * t:2] for i in range(4)])              # [4][c][h+2][t]
    v = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]).astype(np.float32)
    g = w.astype(np.float32)
    u = np.stack([g[..., 0], (g[..., 0] + g[..., 1] + g[..., 2]) * np.float32(0.5),
                  (g[..., 0] - g[..., 1] + g[..., 2]) * np.float32(0.5), g[..., 2]])   # [4][m][c][ky]
    vp, up = split3(v), split3(u)
    order = [(1, 1), (0, 2), (2, 0), (0, 1), (1, 0), (0, 0)]
    m_acc = np.zeros((4, cout, h, t), np.float32)
    lo = np.zeros_like(m_acc)
    for c0 in range(0, cin, 16):
        for ky in range(3):
            for n, (i, j) in enumerate(order):
                part = np.einsum('xmc,xcyt->xmyt', up[i][:, :, c0:c0 + 16, ky].astype(np.float64),
                                 vp[j][:, c0:c0 + 16, ky:ky + h].astype(np.float64))
                if hilo and n < 5:
                    lo = (lo.astype(np.float64) + part).astype(np.float32)
                else:
                    m_acc = (m_acc.astype(np.float64) + part).astype(np.float32)
    if hilo:
        m_acc = m_acc + lo
    out = np.empty((cout, h, 2 * t), np.float32)
    out[:, :, 0::2] = m_acc[0] + m_acc[1] + m_acc[2]
    out[:, :, 1::2] = m_acc[1] - m_acc[2] - m_acc[3]
    return out[:, :, :wd]


def conv_wino2_f32_emulated(x, w):
    cin, h, wd = x.shape
    cout = w.shape[0]
    hp, wp = h + h % 2, wd + wd % 2
    xp = np.pad(x, ((0, 0), (1, 1 + hp - h), (1, 1 + wp - wd))).astype(np.float32)
    bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
    gm = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
    at = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
    ty, tx = hp // 2, wp // 2
    d = np.stack([np.stack([xp[:, i:i + 2 * ty:2, j:j + 2 * tx:2] for j in range(4)]) for i in range(4)])
    v = np.einsum('ai,ijcyx,bj->abcyx', bt, d, bt).astype(np.float32)
    u = np.einsum('ai,mcij,bj->abmc', gm, w.astype(np.float32), gm).astype(np.float32)
    acc = np.zeros((4, 4, cout, ty, tx), np.float32)
    for c0 in range(0, cin, 2):
        acc = (acc.astype(np.float64) + np.einsum('abmc,abcyx->abmyx', u[..., c0:c0 + 2].astype(np.float64),
                                                  v[:, :, c0:c0 + 2].astype(np.float64))).astype(np.float32)
    o = np.einsum('ia,abmyx,jb->mijyx', at, acc, at).astype(np.float32)
    out = np.empty((cout, hp, wp), np.float32)
    for i in range(2):
        for j in range(2):
            out[:, i::2, j::2] = o[:, i, j]
    return out[:, :h, :wd]


def conv_direct_f64(x, w):
    cin, h, wd = x.shape
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    out = np.zeros((w.shape[0], h, wd))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum('mc,cyx->myx', w[:, :, ky, kx].astype(np.float64), xp[:, ky:ky + h, kx:kx + wd])
    return out


def main_conv():
    print()
    for c, hw in ((64, 24), (256, 16), (512, 12)):
        x = np.maximum(rng.standard_normal((c, hw, hw)) * 30 + 5, 0).astype(np.float32)
        w = (rng.standard_normal((64, c, 3, 3)) * np.sqrt(2 / (9 * c))).astype(np.float32)
        ref = conv_direct_f64(x, w)
        print('conv %3d -> 64 @ %dx%d:  bf16x3 1-D Winograd %.2e   with hi / lo accumulators %.2e   '
              'float32 2-D Winograd %.2e' % (c, hw, hw, rel(conv_bf3_emulated(x, w, False), ref),
                                             rel(conv_bf3_emulated(x, w, True), ref),
                                             rel(conv_wino2_f32_emulated(x, w), ref)))


if __name__ == '__main__':
    main_gemm()
    main_conv()
