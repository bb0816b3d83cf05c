#!/usr/bin/env python3
"""Study (CPU): accuracy against MFMA cost of every candidate arithmetic for the head towers' 3x3 256 -> 256 layers
(multitask_headers.py:98-123: conv + ReLU + dropout, four in a row) in the parity mode, whose gate is north_star's 1e-3.

The bf16x3 mode (conv_igemm.hip, SPLIT) spends three bf16 MFMA products per multiplication: x = hi + lo, hi*hi + hi*lo + lo*hi.
Candidates, with their cost in "bf16 MFMA products per direct multiplication" (MI355X dense rates: f16 = bf16, MX-fp8 2x, MX-fp6 /
MX-fp4 4x -- cdna_hip_programming.md section 3):

  direct forms
    bf16                         1      today's throughput mode
    bf16x3                       3      today's parity mode
    f16                          1      one f16 product, no correction
    f16 pair x f16 single        2      activations (hi, lo) f16, weights rounded once
    f16 + bf8 cross              2      hi*hi on the f16 pipe; hi*lo + lo*hi on the MX pipe, all four operands in e5m2 with constant
                                        block scales (lo stored as lo * 2^12)
    f16 + e4m3 cross             2      same with e4m3 and per-32-channel block scales
    f16 + fp6 cross              1.5    e2m3, block scales
    f16 + fp4 cross              1.5    e2m1, block scales
  Winograd F(2x2, 3x3): 16 instead of 36 multiplications per 2x2 tile
    bf16 operands                0.44
    f16 operands                 0.44
    (hi, lo) bf16 operands       1.33
    (hi, lo) f16 operands        1.33

Every candidate is evaluated in float64 ON ITS ROUNDED OPERANDS (the fp32 accumulation of the matrix pipe adds ~1e-7, measured by
winograd_numerics.py), (a) on one layer with exact inputs and (b) on the chain of four layers, each fed with the candidate's own
(re-rounded) output of the layer before, against the float64 chain on unrounded fp32 weights -- what the end-to-end tests see.
Errors are relative to the reference output's RMS: rms | p99.99 | max of |y - ref| / rms(ref), and the tests' own metric
max |y - ref| / (|ref| + rms(ref)).
usage: tower_numerics.py [H W] [--seed s]"""
import sys
import numpy as np
import torch
import torch.nn.functional as F

torch.set_num_threads(8)


# ---------------------------------------------------------------- number formats
def rne_bits(x, drop):                      # float32 -> float32 with `drop` low mantissa bits rounded away (nearest even)
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + ((1 << (drop - 1)) - 1) + ((u >> drop) & 1)) & ~np.uint64((1 << drop) - 1)
    return u.astype(np.uint32).view(np.float32)


def bf16(x):
    return rne_bits(x, 16)


def f16(x):                                 # IEEE half with subnormals FLUSHED to zero (what the kernel stores: the lo part then carries the value)
    y = np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
    y[np.abs(y) < 2.0 ** -14] = 0.0
    return y


def minifloat(x, ebits, mbits, bias, vmax):
    """round-to-nearest-even into a small float format with subnormals, saturating at vmax"""
    x = np.asarray(x, np.float64)
    s = np.sign(x); m = np.abs(x)
    emin = 1 - bias
    e = np.floor(np.log2(np.maximum(m, 1e-300)))
    e = np.maximum(e, emin)
    q = 2.0 ** (e - mbits)
    r = np.round(m / q) * q                 # np.round is half-to-even
    return (s * np.minimum(r, vmax)).astype(np.float64)


FMT = {                                     # ebits, mbits, bias, max, emax (largest power of two)
    "bf8": (5, 2, 15, 57344.0, 15),
    "e4m3": (4, 3, 7, 448.0, 8),
    "fp6": (2, 3, 1, 7.5, 2),               # e2m3
    "fp4": (2, 1, 1, 6.0, 2),               # e2m1
}


def mx_block(x, fmt, axis, const_scale=None):
    """OCP MX quantisation along `axis` in blocks of 32: shared power-of-two scale 2^(floor(log2 max) - emax), elements in `fmt`.
    const_scale: use that exponent for every block instead (the kernel's constant E8M0 operands)."""
    eb, mb, bias, vmax, emax = FMT[fmt]
    x = np.moveaxis(np.asarray(x, np.float64), axis, -1)
    sh = x.shape
    xb = x.reshape(sh[:-1] + (sh[-1] // 32, 32))
    if const_scale is None:
        mx = np.abs(xb).max(-1, keepdims=True)
        se = np.floor(np.log2(np.maximum(mx, 2.0 ** -120))) - emax
    else:
        se = np.full(xb.shape[:-1] + (1,), float(const_scale))
    sc = 2.0 ** se
    y = minifloat(xb / sc, eb, mb, bias, vmax) * sc
    return np.moveaxis(y.reshape(sh), -1, axis)


# ---------------------------------------------------------------- convolutions (float64, torch)
def conv(x, w):
    """x [C,H,W] float64, w [K,C,3,3] float64 -> [K,H,W], SAME zero padding"""
    return F.conv2d(torch.from_numpy(np.ascontiguousarray(x))[None], torch.from_numpy(np.ascontiguousarray(w)), padding=1)[0].numpy()


G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def winograd(x, w, split):
    """F(2x2,3x3) with transformed operands passed through `split` (-> list of (u_part, v_part) product pairs)"""
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
    U = np.einsum("ai,kcij,bj->abkc", G, w, G)                                      # [4,4,K,C]
    tiles = np.stack([np.stack([xp[:, ty:ty + 4, tx:tx + 4] for tx in range(0, W, 2)]) for ty in range(0, H, 2)])   # [Ty,Tx,C,4,4]
    V = np.einsum("ai,yxcij,bj->abyxc", BT, tiles, BT)                             # [4,4,Ty,Tx,C]
    M = 0.0
    for up, vp in split(U.astype(np.float32), V.astype(np.float32)):
        M = M + np.einsum("abyxc,abkc->abyxk", vp, up)
    Y = np.einsum("ia,abyxk,jb->kyixj", AT, M, AT)                                  # [K,Ty,2,Tx,2]
    return Y.reshape(w.shape[0], H, W)


# ---------------------------------------------------------------- candidates: layer(x fp32 [C,H,W], w fp32 [K,C,3,3]) -> float64 output
S_LO = 12                                   # lo parts are stored as lo * 2^12 (f16 hi: |lo| <= 2^-11 |x|)


def c_bf16(x, w):
    return conv(bf16(x).astype(np.float64), bf16(w).astype(np.float64))


def pair(x, rnd):
    hi = rnd(x)
    sc = 2.0 ** 12 if rnd is f16 else 1.0                       # an f16 lo part is stored scaled (it would be subnormal otherwise)
    lo = rnd(((x.astype(np.float64) - hi) * sc).astype(np.float32)) / sc
    return hi.astype(np.float64), lo.astype(np.float64)


def c_bf16x3(x, w):
    xh, xl = pair(x, bf16); wh, wl = pair(w, bf16)
    return conv(xh, wh) + conv(xh, wl) + conv(xl, wh)


def c_f16(x, w):
    return conv(f16(x).astype(np.float64), f16(w).astype(np.float64))


def c_f16_pair_single(x, w):
    xh, xl = pair(x, f16)
    wh = f16(w).astype(np.float64)
    return conv(xh, wh) + conv(xl, wh)


def make_cross(fmt, const):
    def layer(x, w):
        xh = f16(x).astype(np.float64); wh = f16(w).astype(np.float64)
        xl = x.astype(np.float64) - xh; wl = w.astype(np.float64) - wh
        cs = 0 if const else None                                                  # constant block scale 2^0 on the stored values
        q = lambda v, ax, scaled: mx_block(v * (2.0 ** S_LO if scaled else 1.0), fmt, ax, cs) * (2.0 ** -S_LO if scaled else 1.0)
        xh8, xl8 = q(xh, 0, False), q(xl, 0, True)                                 # blocks of 32 channels of a pixel
        wh8, wl8 = q(wh, 1, False), q(wl, 1, True)                                 # blocks of 32 input channels of a (cout, tap)
        return conv(xh, wh) + conv(xh8, wl8) + conv(xl8, wh8)
    return layer


def c_f16_fp6_paired(x, w):
    """The kernel's form (conv_igemm.hip, MXK): an MX block = 16 channels x {hi, lo * 2^11} in e2m3 under ONE shared scale
    2^(floor(log2(max * 16/15)) - 2) (nothing saturates); activations [hi6, lo6'] against weights [lo6', hi6]."""
    xh = f16(x).astype(np.float64); wh = f16(w).astype(np.float64)
    xl = (x.astype(np.float64) - xh) * 2048.0; wl = (w.astype(np.float64) - wh) * 2048.0

    def q(hi, lo, axis):
        hi = np.moveaxis(hi, axis, -1); lo = np.moveaxis(lo, axis, -1)
        sh = hi.shape
        hb = hi.reshape(sh[:-1] + (sh[-1] // 16, 16)); lb = lo.reshape(hb.shape)
        mx = np.maximum(np.abs(hb).max(-1, keepdims=True), np.abs(lb).max(-1, keepdims=True))
        sc = 2.0 ** (np.floor(np.log2(np.maximum(mx * (16.0 / 15.0), 2.0 ** -120))) - 2)
        qh = minifloat(hb / sc, 2, 3, 1, 7.5) * sc; ql = minifloat(lb / sc, 2, 3, 1, 7.5) * sc
        return np.moveaxis(qh.reshape(sh), -1, axis), np.moveaxis(ql.reshape(sh), -1, axis) / 2048.0
    xh6, xl6 = q(xh, xl, 0)
    wh6, wl6 = q(wh, wl, 1)
    return conv(xh, wh) + conv(xh6, wl6) + conv(xl6, wh6)


def make_fp4_paired(lo_shift, blk=16):
    """The fp4 form (conv_igemm.hip, f16mx4): an MX block = `blk` channels x {hi, lo * 2^lo_shift} in e2m1 under ONE shared scale
    2^(floor(log2(max|hi| * 4/3)) - 2) (the largest hi lands in [3, 6]; lo' beyond 6 saturates); activations [hi4, lo4'] against
    weights [lo4', hi4]."""
    f = 2.0 ** lo_shift

    def q(hi, lo, axis):
        hi = np.moveaxis(hi, axis, -1); lo = np.moveaxis(lo, axis, -1)
        sh = hi.shape
        hb = hi.reshape(sh[:-1] + (sh[-1] // blk, blk)); lb = lo.reshape(hb.shape)
        mx = np.abs(hb).max(-1, keepdims=True)
        sc = 2.0 ** (np.floor(np.log2(np.maximum(mx * (4.0 / 3.0), 2.0 ** -120))) - 2)
        qh = minifloat(hb / sc, 2, 1, 1, 6.0) * sc; ql = minifloat(lb / sc, 2, 1, 1, 6.0) * sc
        return np.moveaxis(qh.reshape(sh), -1, axis), np.moveaxis(ql.reshape(sh), -1, axis) / f

    def layer(x, w):
        xh = f16(x).astype(np.float64); wh = f16(w).astype(np.float64)
        xl = (x.astype(np.float64) - xh) * f; wl = (w.astype(np.float64) - wh) * f
        xh4, xl4 = q(xh, xl, 0)
        wh4, wl4 = q(wh, wl, 1)
        return conv(xh, wh) + conv(xh4, wl4) + conv(xl4, wh4)
    return layer


def make_wino(rnd, pairs):
    def split(U, V):
        if not pairs:
            return [(rnd(U).astype(np.float64), rnd(V).astype(np.float64))]
        uh, ul = pair(U, rnd); vh, vl = pair(V, rnd)
        return [(uh, vh), (ul, vh), (uh, vl)]
    return lambda x, w: winograd(x.astype(np.float64), w.astype(np.float64), split)


CANDIDATES = [
    ("bf16 (throughput mode)", 1.0, c_bf16, bf16),
    ("bf16x3 (parity mode, rounds 2-4)", 3.0, c_bf16x3, None),
    ("f16, one product", 1.0, c_f16, f16),
    ("f16 (hi,lo) activations x f16 weights", 2.0, c_f16_pair_single, None),
    ("f16 hi*hi + MX-bf8 cross, constant scales", 2.0, make_cross("bf8", True), None),
    ("f16 hi*hi + MX-e4m3 cross, block scales", 2.0, make_cross("e4m3", False), None),
    ("f16 hi*hi + MX-fp6 (e2m3) cross, block scales", 1.5, make_cross("fp6", False), None),
    ("f16 hi*hi + MX-fp4 (e2m1) cross, block scales", 1.5, make_cross("fp4", False), None),
    ("f16 hi*hi + MX-fp6 cross, {hi, lo*2^11} x 16 ch blocks [built]", 1.5, c_f16_fp6_paired, None),
    ("f16 hi*hi + MX-fp4 cross, {hi, lo*2^11} x 16 ch blocks", 1.5, make_fp4_paired(11), None),
    ("f16 hi*hi + MX-fp4 cross, {hi, lo*2^12} x 16 ch blocks", 1.5, make_fp4_paired(12), None),
    ("f16 hi*hi + MX-fp4 cross, {hi, lo*2^13} x 16 ch blocks", 1.5, make_fp4_paired(13), None),
    ("Winograd F(2x2,3x3), bf16 operands", 16 / 36, make_wino(bf16, False), bf16),
    ("Winograd F(2x2,3x3), f16 operands", 16 / 36, make_wino(f16, False), f16),
    ("Winograd F(2x2,3x3), (hi,lo) bf16 operands", 3 * 16 / 36, make_wino(bf16, True), None),
    ("Winograd F(2x2,3x3), (hi,lo) f16 operands", 3 * 16 / 36, make_wino(f16, True), None),
]


def stats(y, ref):
    rms = np.sqrt((ref ** 2).mean())
    d = np.abs(y - ref)
    return np.sqrt((d ** 2).mean()) / rms, np.quantile(d, 0.9999) / rms, d.max() / rms, (d / (np.abs(ref) + rms)).max()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    H, W = (int(args[0]), int(args[1])) if len(args) >= 2 else (32, 32)
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 0
    C = 256
    rng = np.random.default_rng(seed)
    ws = [(rng.standard_normal((C, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(np.float32) for _ in range(4)]
    masks = [((rng.random((C, H, W)) >= 0.3) / 0.7).astype(np.float32) for _ in range(4)]
    x0 = (np.maximum(rng.standard_normal((C, H, W)), 0) * masks[0]).astype(np.float32)         # a ReLU'd, dropped-out input
    act = lambda y, l: (np.maximum(y, 0) * masks[l]).astype(np.float32)                         # fp32 epilogue: ReLU, dropout scale
    # float64 chain on the unrounded operands
    refs, x = [], x0
    xin = [x0]
    for l in range(4):
        y = conv(x.astype(np.float64), ws[l].astype(np.float64))
        refs.append(y)
        x = act(y, l).astype(np.float64)
        xin.append(x)
    print("tower chain: four 3x3 %d -> %d layers (he-normal), ReLU + dropout 0.3 between them, %dx%d pixels, seed %d" % (C, C, H, W, seed))
    print("errors / rms(reference): rms | p99.99 | max | tests' metric max |d| / (|ref| + rms)")
    print("%-50s %5s  %-41s  %-41s" % ("candidate", "prod", "one layer (exact inputs)", "after four layers (own inputs)"))
    for name, cost, layer, store in CANDIDATES:
        one = stats(layer(xin[1].astype(np.float32), ws[1]), refs[1])
        x = x0
        for l in range(4):
            y = layer(x, ws[l])
            x = act(y, l)
            if store is not None:
                x = store(x)                                   # single-storage modes re-round what the next layer reads
        four = stats(y, refs[3])
        f = lambda s: "%.1e %.1e %.1e %.1e" % s
        print("%-50s %5.2f  %-41s  %-41s" % (name, cost, f(one), f(four)))


if __name__ == "__main__":
    main()
