"""Where the time goes inside the query-side launches of a transformer block (ATTN_Q2P, ATTN_SELF, QFFN, ATTN_P2Q): cycle stamps of
every wave (cutie_amd/csrc/qchain.hip ATL, diagnostic library only), warm (back to back) and cold (a 192 MB copy in front of every
launch, as inside a frame).  Needs the MI355X and the timeline library:

    bash tools/build_diag_attn.sh        # qchain.hip with -DATT_TIMELINE -> tools/abl/libcutie_hip_ATL.so
    CUTIE_AMD_LIB=tools/abl/libcutie_hip_ATL.so python tools/attn_timeline.py [--K 3] [--HW 1620]

Per launch: the hipEvent time of the launch alone (warm / cold), and per stamp id the min / mean / max over waves of the time since
the block's first stamp, in us (cycle counter calibrated against the 100 MHz wall clock of the same stamps).
Stamp ids: see the ATL(n) calls in qchain.hip (0 = kernel entry; the last id = stores issued)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cutie_amd import _lib, ops as O
from cutie_amd.model.weights import pack_linear


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--K', type=int, default=3)
    ap.add_argument('--HW', type=int, default=1620)
    ap.add_argument('--only', default='', help='substring of the launch names to run')
    args = ap.parse_args()
    assert 'ATL' in os.path.basename(_lib.LIB_PATH), 'set CUTIE_AMD_LIB to the timeline library'
    ex = _lib.get_executor()
    g = torch.Generator().manual_seed(0)
    K, HW, Q, C, heads, FF = args.K, args.HW, 16, 256, 8, 2048
    M = K * Q
    dev = 'cuda'
    rn = lambda *s, sc=1.0: (torch.randn(s, generator=g) * sc).to(dev)
    mk = lambda n, kd=C: pack_linear(torch.randn((n, kd), generator=g) / (kd ** 0.5), torch.randn(n, generator=g) * 0.1, dev)
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
    lnp = lambda: ((torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev))
    x, emb = rn(M, C), rn(M, C, sc=0.5)
    lg = rn(K, HW, sc=2.0)
    kvq = rn(K, HW, 3 * C).to(torch.bfloat16)
    Wq, Wo1, Wqkv, Wo2, W1, W2, Wkv = mk(C), mk(C), mk(3 * C), mk(C), mk(FF), mk(C, FF), mk(2 * C)
    ln1, ln2, ln3 = lnp(), lnp(), lnp()
    xn, y, x2 = z(M, C), z(M, C), z(M, C)
    zi = lambda: torch.zeros((M, C), dtype=torch.int64, device=dev)
    a0, a1, a2, a3 = (torch.randn((M, C), generator=g) * 0.1 * 4294967296.0).to(torch.int64).to(dev), zi(), zi(), zi()
    pa = z(K, HW, C, dt=torch.bfloat16)
    launches = []

    def one(name, fn, blocks, stamps=True):
        ol = O.OpList()
        fn(ol)
        tl = torch.zeros((blocks, 16, 16), dtype=torch.int64, device=dev)
        ol.recs[-1][4].extend([0] * (16 - len(ol.recs[-1][4])))
        if stamps:                                         # (p15 is taken in the next-q form of ATTN_P2Q: launch time only)
            ol.recs[-1][4][15] = tl.data_ptr()
        ol.keep.append(tl)
        launches.append((name, ol, tl))

    one('ATTN_Q2P (+parts in, +out-proj)', lambda ol: ol.attn_q2p(None, kvq, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg,
        proj=dict(x=x, W=Wq, emb=emb, ln=ln1, ln_out=xn), acc_in=(a0, W2.bias), out_proj=(Wo1, a1)), heads * K)
    qpre = rn(M, C, sc=0.2)
    one('ATTN_Q2P with q handed in', lambda ol: ol.attn_q2p(None, kvq, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=C, logits=lg,
        q_pre=qpre, out_proj=(Wo1, a1)), heads * K)
    t_ = kvq.cpu()
    kvi = torch.cat([torch.stack([t_[..., :C].reshape(K, HW, heads, 32), t_[..., C:2 * C].reshape(K, HW, heads, 32)], 3).reshape(K, HW, 2 * C), t_[..., 2 * C:]], -1).contiguous().to(dev)
    one('ATTN_Q2P with q handed in, k | v interleaved per head', lambda ol: ol.attn_q2p(None, kvi, None, None, None, K=K, Q=Q, HW=HW, C=C, heads=heads, ldkv=3 * C, voff=32, hstride=64, logits=lg,
        q_pre=qpre, out_proj=(Wo1, a1)), heads * K)
    one('ATTN_SELF (+parts in, +out-proj)', lambda ol: ol.attn_self(None, None, None, K=K, Q=Q, C=C, heads=heads,
        proj=dict(x=xn, W=Wqkv, emb=emb, ln=ln2, ln_out=y), acc_in=(a1, Wo1.bias), out_proj=(Wo2, a2)), heads * K)
    one('QFFN slice 64', lambda ol: ol.qffn(y, x2, a3, rows=M, ln=ln3, W1=W1, W2=W2, acc_in=(a2, Wo2.bias), hid_slice=64), (FF // 64) * K)
    one('QFFN slice 128', lambda ol: ol.qffn(y, x2, a3, rows=M, ln=ln3, W1=W1, W2=W2, acc_in=(a2, Wo2.bias), hid_slice=128), (FF // 128) * K)
    one('ATTN_P2Q (+parts in)', lambda ol: ol.attn_p2q(kvq.view(-1)[2 * C:], None, None, pa, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C,
        proj=dict(x=x2, W=Wkv, emb=emb), acc_in=(a3, W2.bias)), -(-HW // 256) * heads * K)
    qo, xo = z(M, C), z(M, C)
    one('ATTN_P2Q + next q', lambda ol: ol.attn_p2q(kvq.view(-1)[2 * C:], None, None, pa, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C,
        proj=dict(x=x2, W=Wkv, emb=emb), acc_in=(a3, W2.bias), next_q=dict(ln=ln1, W=Wq, q_out=qo, xn_out=xo)), (-(-HW // 256) + 1) * heads * K, stamps=False)

    from cutie_amd.model.weights import linear_as_conv, out_proj_blob
    Woc = linear_as_conv(torch.randn((C, C), generator=g) / 16, torch.randn(C, generator=g) * 0.1, dev)
    blob = out_proj_blob(Woc)
    pixel, pf = rn(K, HW, C).to(torch.bfloat16), z(K, HW, C, dt=torch.bfloat16)
    one('ATTN_P2Q + output projection + residual (flags&32)', lambda ol: ol.attn_p2q(kvq.view(-1)[2 * C:], None, None, pf, K=K, Q=Q, HW=HW, C=C, heads=heads, ldq=3 * C,
        proj=dict(x=x2, W=Wkv, emb=emb), acc_in=(a3, W2.bias), out=dict(Wo=blob, res=pixel)), -(-HW // 64) * K)
    one('the 1x1 conv behind ATTN_P2Q (tile 110)', lambda ol: ol.conv(pa, Woc, pf, B=K, H=1, W=HW, C1=C, ldx1=C, OH=1, OW=HW, ldy=C, res=pixel, ldr=C, tile=110),
        1, stamps=False)
    if args.only:
        launches[:] = [l for l in launches if args.only in l[0]]

    src = torch.zeros(192 << 20, dtype=torch.uint8, device=dev)
    dst = torch.zeros(192 << 20, dtype=torch.uint8, device=dev)
    fl = O.OpList()
    fl.copy2d(src, dst, rows=192, rowbytes=1 << 20, src_stride=1 << 20, dst_stride=1 << 20)
    flush = fl.finalize()
    for _ in range(3):
        ex.run(flush)
    torch.cuda.synchronize()
    t_flush = min(ex.time_ops(flush, 10) for _ in range(3)) * 1e3
    for name, ol, tl in launches:
        arr = ol.finalize()
        for _ in range(3):
            ex.run(arr)
        torch.cuda.synchronize()
        warm = min(ex.time_ops(arr, 30) for _ in range(3)) * 1e3
        seq = np.concatenate([flush, arr])
        cold = min(ex.time_ops(seq, 10) for _ in range(3)) * 1e3 - t_flush
        print(f'\n{name}: launch alone warm {warm:.2f} us, cold {cold:.2f} us  (with stamps)')
        for mode in ('warm', 'cold'):
            acc = []
            for rep in range(5):
                tl.zero_()
                torch.cuda.synchronize()
                if mode == 'cold':
                    ex.run(flush)
                else:
                    ex.run(arr)
                    torch.cuda.synchronize()
                    tl.zero_()
                    torch.cuda.synchronize()
                ex.run(arr)
                torch.cuda.synchronize()
                acc.append(tl.cpu().numpy().astype(np.float64))
            t = acc[-1]
            used = t[:, :, 0] > 0                                              # waves that stamped
            if not used.any():
                continue
            # calibration: cycles per 100 MHz tick over the longest span
            cyc = (np.nanmax(np.where(t[:, :, :14] > 0, t[:, :, :14], np.nan), axis=2) - t[:, :, 0])[used]
            wall = (t[:, :, 15] - t[:, :, 14])[used]
            ok = wall > 0
            mhz = float(np.sum(cyc[ok]) / np.sum(wall[ok]) * 100.0) if ok.any() else float('nan')
            t0 = np.where(used, t[:, :, 0], np.inf).min(axis=1, keepdims=True)  # first entry of the block
            w0 = np.where(used, t[:, :, 14], np.inf).min()
            spread = (np.where(used, t[:, :, 14], np.nan).max() - w0) / 100.0   # us between the first and the last wave entering
            print(f'  {mode}: shader clock ~{mhz:.0f} MHz; waves enter over {spread:.2f} us; last stamp of the grid {float((np.where(used, t[:, :, 15], np.nan)).max() - w0) / 100.0:.2f} us after the first')
            for sid in range(14):
                v = t[:, :, sid]
                m = used & (v > 0)
                if not m.any():
                    continue
                d = ((v - t0) / mhz)[m]
                print(f'    stamp {sid:2d}: min {d.min():6.2f}  mean {d.mean():6.2f}  max {d.max():6.2f} us  ({int(m.sum())} waves)')


if __name__ == '__main__':
    main()
