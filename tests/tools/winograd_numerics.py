#!/usr/bin/env python3
"""Study (CPU, NumPy): what a Winograd F(2x2, 3x3) form of the tower conv would cost in ACCURACY with bf16 operands.
The towers run at the board's power limit, where time follows the MAC count; F(2x2, 3x3) needs 16 instead of 36
multiplications per 2x2 output tile and input channel.  Its operands would be the TRANSFORMED weights (G g G^T) and inputs
(B^T d B) rounded to bf16 -- the input transform adds up to four activations before the rounding, the output transform adds up to
nine products' sums after it.  This script measures, on a layer of the towers' shape (3x3, 256 -> 256, he-normal weights, inputs =
ReLU'd / dropped-out activations), the error of  (a) the direct bf16 conv (what the kernels compute now)  and  (b) the Winograd form
with bf16 transformed operands, fp32 accumulation  against float64, so that the next round can decide with numbers.
usage: winograd_numerics.py [H W]"""
import sys
import numpy as np


def bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32)


G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def direct(x, w, dtype):
    """x [H+2, W+2, C] (zero border), w [3,3,C,K] -> [H, W, K]"""
    H, W = x.shape[0] - 2, x.shape[1] - 2
    out = np.zeros((H, W, w.shape[3]), dtype)
    for ky in range(3):
        for kx in range(3):
            out += x[ky:ky + H, kx:kx + W].astype(dtype).reshape(H * W, -1).dot(w[ky, kx].astype(dtype)).reshape(H, W, -1)
    return out


def winograd(x, w, round_fn, acc):
    H, W = x.shape[0] - 2, x.shape[1] - 2
    U = np.einsum("ai,ijck,bj->abck", G, w.astype(np.float64), G)                  # [4,4,C,K]
    U = round_fn(U.astype(np.float32)).astype(acc)
    out = np.zeros((H, W, w.shape[3]), acc)
    for ty in range(0, H, 2):
        d = x[ty:ty + 4].astype(np.float64)                                        # [4, W+2, C]
        tiles = np.stack([d[:, tx:tx + 4] for tx in range(0, W, 2)])               # [T,4,4,C]
        V = np.einsum("ai,tijc,bj->tabc", BT, tiles, BT)
        V = round_fn(V.astype(np.float32)).astype(acc)
        M = np.einsum("tabc,abck->tabk", V, U)                                     # 16 products per (tile, cout), summed over C in `acc`
        Y = np.einsum("ia,tabk,jb->tijk", AT.astype(acc), M, AT.astype(acc))       # [T,2,2,K]
        out[ty:ty + 2] = Y.transpose(1, 0, 2, 3).reshape(2, W, -1)
    return out


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 32)
    C = K = 256
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((3, 3, C, K)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    a = np.maximum(rng.standard_normal((H, W, C)), 0).astype(np.float32)           # ReLU'd activations
    a *= (rng.random((H, W, C)) >= 0.3) / 0.7                                      # dropout, scaled
    x = np.zeros((H + 2, W + 2, C), np.float32); x[1:-1, 1:-1] = bf16(a)           # the stored bf16 activations
    wb = bf16(w)
    ref = direct(x, wb, np.float64)
    rms = np.sqrt((ref ** 2).mean())
    rel = lambda y: (np.sqrt(((y - ref) ** 2).mean()) / rms, np.abs(y - ref).max() / rms)
    print("layer 3x3 %d -> %d on %dx%d, bf16-stored inputs / weights; errors against float64 of the SAME bf16 operands, relative to the output RMS" % (C, K, H, W))
    print("  direct, fp32 accumulate (the kernels now)            rms %.2e  max %.2e" % rel(direct(x, wb, np.float32)))
    print("  Winograd F(2x2,3x3), fp32 transforms + fp32 operands  rms %.2e  max %.2e" % rel(winograd(x, wb, lambda v: v, np.float32)))
    print("  Winograd F(2x2,3x3), transformed operands in bf16     rms %.2e  max %.2e" % rel(winograd(x, wb, bf16, np.float32)))
    y = winograd(x, wb, bf16, np.float32)
    print("  (for scale: rounding the direct conv's OUTPUT to bf16 costs rms %.2e)" % rel(bf16(ref.astype(np.float32)).astype(np.float64))[0])


if __name__ == "__main__":
    main()
