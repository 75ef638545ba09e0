"""Numerics of the two-piece fp16 split inside a 3x3 convolution (csrc/conv_h2.hip).

An operand a is written as hi + lo with hi = fp16(s a), lo = fp16(s a - hi) (round to nearest even;
the residual is exact in float32): 2 x 11 significand bits.  A product of two such operands is
hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16 (every fp16 x fp16 product exact in float32, float32
accumulation); lo lo (2^-22 relative) is dropped.  s is a power of two:
  * filters: fixed when the bank is packed, so that max |U| s lies in [2^13, 2^14);
  * inputs: from the blob's max |x| (tracked on the device by the kernel that wrote it), so that
    max |x| s lies in [2^13, 2^14) -- the 1-D Winograd input transform adds two values, |V| < 2^15.
Nothing overflows by construction; what is small against the maximum loses relative precision only
below 2^-17 of it (fp16 subnormals: absolute spacing 2^-24 of 2^14), further down if the matrix
cores flush fp16 subnormals (`flush=True`: everything below 2^-14 becomes zero).

Emulated in numpy: 1-D Winograd F(2,3) along x (V = B^T d, U = G g in float32), pieces in fp16,
products accumulated in float64 and rounded to float32 once per MFMA (16 channels x one kernel row x
one product term), in the kernel's order.  Reference: float64 direct convolution.  Printed: max |err|
/ max |ref|, next to the float32 2-D Winograd form the engine ships for the same layer.

    python tools/f16x2_numerics.py
"""
import numpy as np

from bf16x3_numerics import conv_direct_f64, conv_wino2_f32_emulated, conv_bf3_emulated, rel


def pow2_scale(absmax):
    """power of two s with absmax * s in [2^13, 2^14) (what the kernel derives from the exponent)"""
    if absmax == 0 or not np.isfinite(absmax):
        return np.float32(1)
    e = int(np.floor(np.log2(float(absmax))))
    return np.float32(2.0 ** (13 - e))


def split2(x, flush):
    """float32 -> (hi, lo) as float32 values of fp16 numbers"""
    x = x.astype(np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    hi, lo = hi.astype(np.float32), lo.astype(np.float32)
    if flush:
        hi = np.where(np.abs(hi) < 2.0 ** -14, 0, hi).astype(np.float32)
        lo = np.where(np.abs(lo) < 2.0 ** -14, 0, lo).astype(np.float32)
    return hi, lo


def conv_h2_emulated(x, w, flush=False, x_absmax=None):
    cin, h, wd = x.shape
    cout = w.shape[0]
    s_v = pow2_scale(np.abs(x).max() if x_absmax is None else x_absmax)
    xp = np.pad(x * s_v, ((0, 0), (1, 1), (1, 1 + wd % 2))).astype(np.float32)
    t = (wd + 1) // 2
    d = np.stack([xp[:, :, i:i + 2 * t:2] for i in range(4)])              # [4][c][h+2][t]
    v = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]).astype(np.float32)
    g = w.astype(np.float32)
    u = np.stack([g[..., 0], (g[..., 0] + g[..., 1] + g[..., 2]) * np.float32(0.5),
                  (g[..., 0] - g[..., 1] + g[..., 2]) * np.float32(0.5), g[..., 2]])   # [4][m][c][ky]
    s_w = pow2_scale(np.abs(u).max())
    u = (u * s_w).astype(np.float32)
    assert np.abs(v).max() < 65504 and np.abs(u).max() < 65504
    vp, up = split2(v, flush), split2(u, flush)
    order = [(1, 0), (0, 1), (0, 0)]            # (U piece, V piece): lo hi, hi lo, hi hi
    acc = np.zeros((4, cout, h, t), np.float32)
    for c0 in range(0, cin, 16):
        for ky in range(3):
            for i, j in order:
                part = np.einsum('xmc,xcyt->xmyt', up[i][:, :, c0:c0 + 16, ky].astype(np.float64),
                                 vp[j][:, c0:c0 + 16, ky:ky + h].astype(np.float64))
                acc = (acc.astype(np.float64) + part).astype(np.float32)
    acc = acc * np.float32(1.0 / (float(s_v) * float(s_w)))
    out = np.empty((cout, h, 2 * t), np.float32)
    out[:, :, 0::2] = acc[0] + acc[1] + acc[2]
    out[:, :, 1::2] = acc[1] - acc[2] - acc[3]
    return out[:, :, :wd]


BT43 = np.float32([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                   [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]])
G43 = np.float32([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                  [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
AT43 = np.float32([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]])


def conv_h2_f43_emulated(x, w):
    """The same kernel with 1-D Winograd F(4,3) along x instead of F(2,3): six components per four outputs
    (18/36 of a direct convolution's multiplies against 24/36: a quarter fewer MFMAs -- VERDICT r5 item 3's
    third candidate, priced here before anything is built).  Transforms in float32 (each row of B^T / G / A^T
    as one float32 expression, left to right), pieces in fp16, three products, float32 accumulation per MFMA."""
    cin, h, wd = x.shape
    cout = w.shape[0]
    t = (wd + 3) // 4
    xp = np.pad(x, ((0, 0), (1, 1), (1, 4 * t + 1 - wd))).astype(np.float32)
    d = np.stack([xp[:, :, i:i + 4 * t:4] for i in range(6)])                  # [6][c][h+2][t]
    v = np.zeros((6,) + d.shape[1:], np.float32)
    for r in range(6):
        for k in range(6):
            if BT43[r, k]:
                v[r] = (v[r] + BT43[r, k] * d[k]).astype(np.float32)
    g = w.astype(np.float32)
    u = np.zeros((6,) + g.shape[:3], np.float32)                               # [6][m][c][ky]
    for r in range(6):
        for k in range(3):
            if G43[r, k]:
                u[r] = (u[r] + G43[r, k] * g[..., k]).astype(np.float32)
    s_v, s_w = pow2_scale(np.abs(v).max()), pow2_scale(np.abs(u).max())
    v, u = (v * s_v).astype(np.float32), (u * s_w).astype(np.float32)
    vp, up = split2(v, False), split2(u, False)
    acc = np.zeros((6, cout, h, t), np.float32)
    for c0 in range(0, cin, 16):
        for ky in range(3):
            for i, j in [(1, 0), (0, 1), (0, 0)]:
                part = np.einsum('xmc,xcyt->xmyt', up[i][:, :, c0:c0 + 16, ky].astype(np.float64),
                                 vp[j][:, c0:c0 + 16, ky:ky + h].astype(np.float64))
                acc = (acc.astype(np.float64) + part).astype(np.float32)
    acc = acc * np.float32(1.0 / (float(s_v) * float(s_w)))
    out = np.zeros((cout, h, 4 * t), np.float32)
    for r in range(4):
        o = np.zeros((cout, h, t), np.float32)
        for k in range(6):
            if AT43[r, k]:
                o = (o + AT43[r, k] * acc[k]).astype(np.float32)
        out[:, :, r::4] = o
    return out[:, :, :wd]


def main():
    rng = np.random.RandomState(0)
    print('one layer, 64 output channels; "act" = post-ReLU inputs max(30 N + 5, 0), "grad" = heavy-tailed\n'
          'signed inputs 0.1 N exp(1.5 N) (times the stated factor); He-scaled filters')
    for c, hw in ((64, 24), (256, 16), (512, 12)):
        w = (rng.standard_normal((64, c, 3, 3)) * np.sqrt(2 / (9 * c))).astype(np.float32)
        for kind, x in (('act', np.maximum(rng.standard_normal((c, hw, hw)) * 30 + 5, 0)),
                        ('grad', 0.1 * rng.standard_normal((c, hw, hw)) * np.exp(1.5 * rng.standard_normal((c, hw, hw)))),
                        ('grad x 1e-9', 1e-10 * rng.standard_normal((c, hw, hw)) * np.exp(1.5 * rng.standard_normal((c, hw, hw)))),
                        ('act x 1e4', 1e4 * np.maximum(rng.standard_normal((c, hw, hw)) * 30 + 5, 0))):
            x = x.astype(np.float32)
            ref = conv_direct_f64(x, w)
            print('conv %3d -> 64 @ %2dx%-2d %-12s fp16x2 %.2e   subnormals flushed %.2e   absmax bound 2^6 loose '
                  '%.2e / flushed %.2e   | f32 2-D Winograd %.2e   bf16x3 %.2e'
                  % (c, hw, hw, kind, rel(conv_h2_emulated(x, w), ref), rel(conv_h2_emulated(x, w, True), ref),
                     rel(conv_h2_emulated(x, w, False, np.abs(x).max() * 64), ref),
                     rel(conv_h2_emulated(x, w, True, np.abs(x).max() * 64), ref),
                     rel(conv_wino2_f32_emulated(x, w), ref), rel(conv_bf3_emulated(x, w, False), ref)))
            print('%40s the same with 1-D F(4,3): %.2e' % ('', rel(conv_h2_f43_emulated(x, w), ref)))
    print('\nfour layers 256 -> 256 @ 16x16 with ReLU between them, every layer fed the previous one\'s own output')
    c, hw = 256, 16
    x0 = np.maximum(rng.standard_normal((c, hw, hw)) * 30 + 5, 0).astype(np.float32)
    ws = [(rng.standard_normal((c, c, 3, 3)) * np.sqrt(2 / (9 * c))).astype(np.float32) for _ in range(4)]
    ref = x0.astype(np.float64)
    a = b = f43 = x0
    for n, w in enumerate(ws):
        ref = np.maximum(conv_direct_f64(ref, w), 0)
        a = np.maximum(conv_h2_emulated(a, w), 0)
        b = np.maximum(conv_wino2_f32_emulated(b, w), 0)
        f43 = np.maximum(conv_h2_f43_emulated(f43, w), 0)
        print('after layer %d: fp16x2 %.2e   f32 2-D Winograd %.2e   fp16x2 with 1-D F(4,3) %.2e'
              % (n + 1, rel(a, ref), rel(b, ref), rel(f43, ref)))


if __name__ == '__main__':
    main()
