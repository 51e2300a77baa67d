"""CPU emulation of the INTEGER logic of conv_bufload_kernel (cutie_amd/csrc/conv_bufload.hip): per-thread chunk offsets, tap
validity masks, the wave-uniform tap / channel / source state and its advance, the shifted resource base, out-of-bounds zero
fill, weight offsets, and the loop-invariant LDS indices -- checked against torch's conv2d and against the index formulas of
conv_igemm_kernel.  The MFMA fragment layout, swizzle and epilogue are shared with the validated kernel and not emulated.
Written because the kernel was added when no GPU time was left: this is the part of it that can be verified without one.

    python tools/emulate_bufload_addressing.py"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F


def swz(cpr, row):
    return ((row >> 3) & 1) * 3 if cpr == 4 else (row & (cpr - 1))


def emulate(B, H, W, C1, C2, Cout, k, stride, pad, BM, BN, BK, NT, ldx1=None, ldx2=None, seed=0):
    g = torch.Generator().manual_seed(seed)
    Cin = C1 + C2
    ldx1 = ldx1 or C1
    ldx2 = ldx2 or max(C2, 8)
    x1 = torch.randn(B, H, W, ldx1, generator=g)
    x2 = torch.randn(B, H, W, ldx2, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    M, OHW = B * OH * OW, OH * OW
    K = k * k * Cin
    Kpad = -(-K // 128) * 128
    coutpad = -(-Cout // 128) * 128
    wp = torch.zeros(coutpad, Kpad)
    wp[:Cout, :K] = w.permute(0, 2, 3, 1).reshape(Cout, K)
    assert Cin % BK == 0 and (C2 == 0 or C1 % BK == 0) and Kpad % BK == 0
    # flat "memory": descriptors address BYTES relative to a (shifted) base; emulate with element indices into flat arrays
    f1, f2, fw = x1.reshape(-1), (x2.reshape(-1) if C2 else None), wp.reshape(-1)
    shift = pad * W + pad
    base1, base2 = -shift * ldx1, -shift * ldx2                 # element index of the shifted base inside f1 / f2
    OOB = 0x80000000
    CPR, RPT = BK // 8, NT // (BK // 8)
    NX, NWC = BM * CPR // NT, BN * CPR // NT
    nk = Kpad // BK
    out = torch.zeros(M, Cout)
    gy = -(-Cout // BN)
    for m0 in range(0, M, BM):
        for n0 in range(0, gy * BN, BN):
            acc = torch.zeros(BM, BN)
            # per-thread invariant state
            xoff1 = np.zeros((NT, NX), np.int64); xoff2 = np.zeros((NT, NX), np.int64); vmask = np.zeros((NT, NX), np.int64)
            woff = np.zeros((NT, NWC), np.int64)
            for tid in range(NT):
                kc, trow = tid % CPR, tid // CPR
                for i in range(NX):
                    m = m0 + trow + i * RPT
                    valid = m < M
                    mm = m if valid else 0
                    b, rem = divmod(mm, OHW)
                    oh, ow = divmod(rem, OW)
                    ih0, iw0 = oh * stride, ow * stride
                    pix = (b * H + ih0) * W + iw0
                    xoff1[tid, i] = pix * ldx1 * 2 + kc * 16
                    xoff2[tid, i] = pix * ldx2 * 2 + kc * 16
                    mk, t = 0, 0
                    for kh in range(k):
                        for kw in range(k):
                            ih, iw = ih0 - pad + kh, iw0 - pad + kw
                            if valid and 0 <= ih < H and 0 <= iw < W:
                                mk |= 1 << t
                            t += 1
                    vmask[tid, i] = mk
                for i in range(NWC):
                    woff[tid, i] = (n0 + trow + i * RPT) * Kpad * 2 + kc * 16
            assert xoff1.max() < 2 ** 31 and woff.max() < 2 ** 31
            tap = kh = kw = c0 = 0
            wsoff = 0
            for ks in range(nk):
                xt, wt = torch.zeros(BM, BK), torch.zeros(BN, BK)
                in1 = c0 < C1
                ldx = ldx1 if in1 else ldx2
                soff = ((kh * W + kw) * ldx + (c0 if in1 else c0 - C1)) * 2
                assert soff >= 0
                tapbit = 1 << (tap & 31)
                flat, base = (f1, base1) if in1 else (f2, base2)
                for tid in range(NT):
                    kc, trow = tid % CPR, tid // CPR
                    for i in range(NX):
                        vo = (xoff1 if in1 else xoff2)[tid, i] if (vmask[tid, i] & tapbit) else OOB
                        row = trow + i * RPT
                        if vo < 0x7fffffff:
                            e = base + (vo + soff) // 2
                            assert 0 <= e and e + 8 <= flat.numel(), 'a valid chunk must lie inside the tensor'
                            xt[row, kc * 8:kc * 8 + 8] = flat[e:e + 8]
                    for i in range(NWC):
                        e = (woff[tid, i] + wsoff) // 2
                        wt[trow + i * RPT, kc * 8:kc * 8 + 8] = fw[e:e + 8]
                acc += xt @ wt.t()
                wsoff += BK * 2
                wrap = c0 + BK >= Cin
                c0 = 0 if wrap else c0 + BK
                tap += wrap; kw += wrap
                wrap2 = kw == k
                kw = 0 if wrap2 else kw
                kh += wrap2
            mm = min(BM, M - m0)
            nn = min(BN, Cout - n0)
            if nn > 0:
                out[m0:m0 + mm, n0:n0 + nn] = acc[:mm, :nn]
    xin = x1[..., :C1] if not C2 else torch.cat([x1[..., :C1], x2[..., :C2]], -1)
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w, None, stride, pad).permute(0, 2, 3, 1).reshape(M, Cout)
    return float((out - ref).abs().max()), float(ref.abs().max())


def check_lds_indices():
    for BM, BN, WM, WN, BK in [(128, 64, 2, 2, 64), (64, 64, 2, 2, 64), (64, 128, 2, 2, 64), (32, 64, 2, 2, 128), (64, 64, 2, 2, 128),
                               (128, 128, 2, 4, 64), (32, 64, 2, 2, 64)]:
        NT, CPR = WM * WN * 64, BK // 8
        RPT, NX, NWC = NT // CPR, BM * CPR // NT, BN * CPR // NT
        TM, TN, KSUB = BM // WM // 16, BN // WN // 16, BK // 32
        for tid in range(NT):
            kc, trow = tid % CPR, tid // CPR
            wrx = trow * CPR + (kc ^ swz(CPR, trow))
            for i in range(NX):
                row = trow + i * RPT
                assert wrx + i * RPT * CPR == row * CPR + (kc ^ swz(CPR, row))
            for i in range(NWC):
                n = trow + i * RPT
                assert wrx + (BM + i * RPT) * CPR == (BM + n) * CPR + (kc ^ swz(CPR, n))
            lane, wave = tid & 63, tid >> 6
            wm, wn = wave // WN, wave % WN
            pm0, cn0, l15, l4 = wm * (BM // WM), wn * (BN // WN), lane & 15, lane >> 4
            for j in range(KSUB):
                rdx = (pm0 + l15) * CPR + ((j * 4 + l4) ^ swz(CPR, pm0 + l15))
                rdw = (BM + cn0 + l15) * CPR + ((j * 4 + l4) ^ swz(CPR, cn0 + l15))
                for t in range(TM):
                    row = pm0 + t * 16 + l15
                    assert rdx + t * 16 * CPR == row * CPR + ((j * 4 + l4) ^ swz(CPR, row))
                for t in range(TN):
                    row = cn0 + t * 16 + l15
                    assert rdw + t * 16 * CPR == (BM + row) * CPR + ((j * 4 + l4) ^ swz(CPR, row))
    print('LDS indices: loop-invariant forms equal the per-access formulas of conv_igemm_kernel for all 7 tiles')


if __name__ == '__main__':
    check_lds_indices()
    cases = [
        dict(B=2, H=9, W=11, C1=64, C2=0, Cout=40, k=3, stride=1, pad=1),
        dict(B=1, H=10, W=7, C1=128, C2=0, Cout=64, k=1, stride=1, pad=0),
        dict(B=2, H=9, W=12, C1=64, C2=0, Cout=72, k=3, stride=2, pad=1),
        dict(B=1, H=8, W=9, C1=64, C2=0, Cout=130, k=1, stride=2, pad=0),
        dict(B=2, H=6, W=7, C1=64, C2=64, Cout=96, k=3, stride=1, pad=1, ldx1=80, ldx2=72),
        dict(B=1, H=7, W=6, C1=128, C2=64, Cout=64, k=1, stride=1, pad=0),
    ]
    for c, (BM, BN, BK) in itertools.product(cases, [(128, 64, 64), (32, 64, 64), (64, 128, 64)]):
        err, scale = emulate(**c, BM=BM, BN=BN, BK=BK, NT=256)
        print(f'{c}  tile {BM}x{BN}x{BK}: max err {err:.2e} (scale {scale:.2f})')
        assert err < 1e-4 * max(1.0, scale)
    err, scale = emulate(B=1, H=6, W=6, C1=128, C2=128, Cout=64, k=3, stride=1, pad=1, BM=32, BN=64, BK=128, NT=256)
    print('BK=128 two-source', err)
    assert err < 1e-4 * max(1.0, scale)
    print('addressing emulation: all cases agree with conv2d')
