"""Torch-CPU interpreter of the ``cutie_op`` descriptors (include/cutie_hip.h).  TEST INFRASTRUCTURE ONLY.

Two uses:
  * ``-m "not gpu"`` tests inject it with ``cutie_amd._lib.set_executor_for_testing`` so the whole host side
    (plan builders, memory bank, InferenceCore mirror) is checked against the oracle without a GPU;
  * ``-m gpu`` kernel unit tests run the same descriptor through the HIP kernel (device buffers) and through
    this interpreter (host copies) and compare.
It reads/writes host memory by raw address, exactly like the kernels do with device pointers.  Values stored
as bf16 are rounded to bf16 here too (fp32 accumulation inside an op), so HIP-vs-mock differences are
accumulation-order noise only.
"""
import ctypes
import math
import numpy as np
import torch
import torch.nn.functional as F

from cutie_amd import ops as O

BF16, F32, I32, U8, U64, I64 = 'bf16', 'f32', 'i32', 'u8', 'u64', 'i64'
_NP = {BF16: (np.uint16, 2), F32: (np.float32, 4), I32: (np.int32, 4), U8: (np.uint8, 1), U64: (np.uint64, 8), I64: (np.int64, 8)}


def view(ptr, dtype, shape, strides=None):
    """Tensor aliasing host memory at `ptr` (strides in elements, default contiguous)."""
    npdt, isz = _NP[dtype]
    shape = tuple(int(s) for s in shape)
    if strides is None:
        strides, acc = [], 1
        for s in reversed(shape):
            strides.append(acc)
            acc *= s
        strides = tuple(reversed(strides))
    extent = 1 + sum((s - 1) * st for s, st in zip(shape, strides)) if all(s > 0 for s in shape) else 0
    if extent == 0:
        return torch.zeros(shape)
    buf = (ctypes.c_char * (extent * isz)).from_address(int(ptr))
    a = np.frombuffer(buf, dtype=npdt)
    t = torch.from_numpy(a)
    if dtype == BF16:
        t = t.view(torch.bfloat16)
    elif dtype == U64:
        t = t.view(torch.int64)
    return t.as_strided(shape, strides)


def _act(v, code):
    if code == O.ACT_RELU:
        return F.relu(v)
    if code == O.ACT_SIGMOID:
        return torch.sigmoid(v)
    if code == O.ACT_SQ1:
        return v * v + 1
    return v


def _clamp_logit(p):
    p = p.clamp(1e-7, 1 - 1e-7)
    return torch.log(p / (1 - p))


def _up_coords(n_out, n_in, scale):
    src = ((torch.arange(n_out, dtype=torch.float32) + 0.5) * scale - 0.5).clamp(min=0)
    i0 = src.floor().long().clamp(max=n_in - 1)
    i1 = (i0 + 1).clamp(max=n_in - 1)
    return i0, i1, src - i0.float()


def _bilinear(x, scale_inv):
    """x [..., h, w] fp32 -> upsampled by 1/scale_inv, align_corners=False (same formula as the kernels)."""
    h, w = x.shape[-2:]
    oh, ow = int(round(h / scale_inv)), int(round(w / scale_inv))
    y0, y1, ly = _up_coords(oh, h, scale_inv)
    x0, x1, lx = _up_coords(ow, w, scale_inv)
    ly = ly.view(-1, 1)
    lx = lx.view(1, -1)
    g = lambda yy, xx: x[..., yy, :][..., :, xx]
    return ((1 - ly) * (1 - lx)) * g(y0, x0) + ((1 - ly) * lx) * g(y0, x1) + (ly * (1 - lx)) * g(y1, x0) + (ly * lx) * g(y1, x1)


def unpack_conv_weight(wp, cout, kh, kw, cin):
    """packed bf16 [CoutPad, Kpad] -> fp32 [cout, cin, kh, kw]"""
    return wp[:cout, :kh * kw * cin].float().view(cout, kh, kw, cin).permute(0, 3, 1, 2).contiguous()


class MockExecutor:
    is_mock = True

    def run(self, arr):
        for rec in arr:
            self.run_one(rec)

    def stream(self):
        return 0

    def run_one(self, rec):
        kind = int(rec['kind'])
        fn = getattr(self, '_op_%d' % kind, None)
        if fn is None:
            raise NotImplementedError(f'mock op kind {kind}')
        fn(int(rec['flags']), [int(v) for v in rec['i']], [float(v) for v in rec['f']], [int(v) for v in rec['p']])

    # ---- CONV ---------------------------------------------------------------------
    def _op_1(self, flags, i, f, p):
        B, H, W, C1, C2, ldx1, ldx2, OH, OW, Cout, ldy, KH, KW, stride, pad, ldr, Kpad, tile = i[:18]
        Cin = C1 + C2
        x = view(p[0], BF16, (B, H, W, C1), (H * W * ldx1, W * ldx1, ldx1, 1)).float()
        if C2:
            x2 = view(p[1], BF16, (B, H, W, C2), (H * W * ldx2, W * ldx2, ldx2, 1)).float()
            x = torch.cat([x, x2], -1)
        if flags & O.F_RELU_IN:
            x = F.relu(x)
        coutpad = -(-Cout // 128) * 128
        w = unpack_conv_weight(view(p[2], BF16, (coutpad, Kpad)), Cout, KH, KW, Cin)
        bias = view(p[3], F32, (Cout,)).clone() if p[3] else None
        if getattr(self, 'per_sample_conv', False):      # batch-invariant rounding (torch's CPU conv may sum differently per batch size)
            y = torch.cat([F.conv2d(x[b:b + 1].permute(0, 3, 1, 2), w, bias, stride, pad) for b in range(B)]).permute(0, 2, 3, 1)
        else:
            y = F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride, pad).permute(0, 2, 3, 1)
        assert y.shape[1] == OH and y.shape[2] == OW, (y.shape, OH, OW)
        if p[4]:
            if (flags & O.F_RES_BCAST) and f[0] > 0:                    # clips in lock step: one broadcast residual per group of f0 objects, f1 rows apart
                kg, gs = int(f[0]), int(f[1])
                r = view(p[4], BF16, (B // kg, OH, OW, Cout), (gs * ldr, OW * ldr, ldr, 1)).float()
                y = y + r.repeat_interleave(kg, 0)
            else:
                rb = 1 if flags & O.F_RES_BCAST else B
                y = y + view(p[4], BF16, (rb, OH, OW, Cout), (OH * OW * ldr, OW * ldr, ldr, 1)).float()
        y = _act(y, (flags >> O.ACT_SHIFT) & 7)
        out = view(p[5], F32 if flags & O.F_OUT_F32 else BF16, (B, OH, OW, Cout), (OH * OW * ldy, OW * ldy, ldy, 1))
        out.copy_(y)
        if len(p) > 8 and p[8]:                                         # side job: clear the next conv's GAP accumulator
            view(p[8], I64, (i[21],)).zero_()
        if len(p) > 7 and p[7]:                                         # GAP accumulation: fixed-point sums of the STORED values
            stored = out.float().reshape(B, OH * OW, Cout)
            view(p[7], I64, (B, Cout)).add_(torch.round(stored.double().sum(1) * 16777216.0).to(torch.int64))

    # ---- MAXPOOL --------------------------------------------------------------------
    def _op_2(self, flags, i, f, p):
        B, H, W, C, OH, OW = i[:6]
        x = view(p[0], BF16, (B, H, W, C)).float().permute(0, 3, 1, 2)
        y = F.max_pool2d(x, 3, 2, 1)
        if flags & 1:
            y = F.relu(y)
        view(p[1], BF16, (B, OH, OW, C)).copy_(y.permute(0, 2, 3, 1))

    # ---- IMG_PREP -------------------------------------------------------------------
    def _op_3(self, flags, i, f, p):
        h0, w0, H, W, pl, pt, K = i[:7]
        K = K if p[1] else 1
        img = view(p[0], F32, (3, h0, w0))
        full = torch.zeros(3, H, W)
        full[:, pt:pt + h0, pl:pl + w0] = img
        mean = torch.tensor(f[0:3]).view(3, 1, 1)
        std = torch.tensor(f[3:6]).view(3, 1, 1)
        full = (full - mean) / std
        out = torch.zeros(K, H, W, 8)
        out[..., 0:3] = full.permute(1, 2, 0)
        if p[1]:
            m = view(p[1], F32, (K, H, W))
            out[..., 3] = m
            out[..., 4] = (m.sum(0, keepdim=True) - m).clamp(0, 1)
        view(p[2], BF16, (K, H, W, 8)).copy_(out)

    # ---- UPSAMPLE2X_ADD ---------------------------------------------------------------
    def _op_4(self, flags, i, f, p):
        B, h, w, C = i[:4]
        g = view(p[0], BF16, (B, h, w, C)).float().permute(0, 3, 1, 2)
        if len(i) > 4 and i[4] > 0:                                     # clips in lock step: one skip map per group of i4 objects, i5 pixels apart
            skip = view(p[1], BF16, (B // i[4], 2 * h, 2 * w, C), (i[5] * C, 2 * w * C, C, 1)).float().repeat_interleave(i[4], 0)
        else:
            skip = view(p[1], BF16, (1, 2 * h, 2 * w, C)).float()
        y = _bilinear(g, 0.5).permute(0, 2, 3, 1) + skip
        view(p[2], BF16, (B, 2 * h, 2 * w, C)).copy_(y)

    # ---- AREA_DOWN -------------------------------------------------------------------
    def _op_5(self, flags, i, f, p):
        B, H, W, C, ldx, ldy, r, Cz = i[:8]
        x = view(p[0], F32 if flags & 1 else BF16, (B, H, W, C), (H * W * ldx, W * ldx, ldx, 1)).float()
        y = F.avg_pool2d(x.permute(0, 3, 1, 2), r).permute(0, 2, 3, 1)
        oh, ow = H // r, W // r
        out = view(p[1], BF16, (B, oh, ow, max(C, Cz)), (oh * ow * ldy, ow * ldy, ldy, 1))
        out[..., :C] = y.to(torch.bfloat16)
        if (flags & 1) and Cz > C:
            out[..., C:Cz] = 0

    def _op_39(self, flags, i, f, p):                                   # AREA_DOWN3 = three AREA_DOWNs
        for q in range(3):
            self._op_5((flags >> q) & 1, i[8 * q:8 * q + 8], f, p[2 * q:2 * q + 2])

    # ---- MASK_DOWN --------------------------------------------------------------------
    def _op_6(self, flags, i, f, p):
        K, H, W, r = i[:4]
        m = view(p[0], F32, (K, H, W))
        m16 = F.avg_pool2d(m.unsqueeze(0), r)[0]
        h, w = H // r, W // r
        view(p[2], F32, (K, h, w)).copy_(m16)
        out = torch.zeros(K, h, w, 8)
        out[..., 0] = m16
        out[..., 1] = (m16.sum(0, keepdim=True) - m16).clamp(0, 1)
        ld = i[4] if len(i) > 4 and i[4] > 8 else 8
        view(p[1], BF16, (K, h, w, 8), (h * w * ld, w * ld, ld, 1)).copy_(out)

    # ---- GAP / ECA ----------------------------------------------------------------------
    def _op_7(self, flags, i, f, p):
        B, HW, C = i[:3]
        x = view(p[0], BF16, (B, HW, C)).float()
        nchunk = -(-HW // 64)
        part = view(p[2], F32, (B, nchunk, C))
        for k in range(nchunk):
            part[:, k] = x[:, k * 64:(k + 1) * 64].sum(1)
        if not (flags & 1):
            view(p[1], F32, (B, C)).copy_(part.sum(1) / HW)

    def _op_8(self, flags, i, f, p):
        B, HW, C = i[:3]
        x = view(p[0], BF16, (B, HW, C)).float()
        nchunk = -(-HW // 64)
        gap = view(p[1], F32, (B, C))
        if flags & 1:                                                   # fixed-point sums from the producing conv
            gap.copy_((view(p[5], I64, (B, C)).double() / 16777216.0).float() / HW)
        else:
            gap.copy_(view(p[5], F32, (B, nchunk, C)).sum(1) / HW)
        wk = view(p[2], F32, (5,))
        sc = torch.sigmoid(F.conv1d(gap.view(B, 1, C), wk.view(1, 1, 5), None, 1, 2)).view(B, 1, C)
        r = view(p[3], BF16, (B, HW, C)).float()
        out = view(p[4], BF16, (B, HW, C))
        out.copy_(x * sc + r)
        if len(p) > 8 and p[6]:                                         # fused Cout = 1 head on the stored output
            w = view(p[6], BF16, (C,)).float()
            bias = view(p[7], F32, (1,)) if p[7] else torch.zeros(1)
            view(p[8], F32, (B, HW)).copy_((F.relu(out.float()) * w).sum(-1) + bias)

    # ---- GRU ------------------------------------------------------------------------------
    def _op_9(self, flags, i, f, p):
        n, C = i[:2]
        v = view(p[0], F32, (n, 3 * C))
        h = view(p[1], F32, (n, C))
        fg, u, nv = torch.sigmoid(v[:, :C]), torch.sigmoid(v[:, C:2 * C]), torch.tanh(v[:, 2 * C:])
        nh = fg * h * (1 - u) + u * nv
        h.copy_(nh)
        view(p[2], BF16, (n, C)).copy_(nh)

    # ---- SEG_AGG / UP4_SOFTMAX ---------------------------------------------------------------
    def _op_10(self, flags, i, f, p):
        K, hw = i[:2]
        pr = torch.sigmoid(view(p[0], F32, (K, hw)))
        agg = view(p[1], F32, (K + 1, hw))
        agg[0] = _clamp_logit(torch.prod(1 - pr, dim=0))
        agg[1:] = _clamp_logit(pr)

    def _op_11(self, flags, i, f, p):
        P, h, w = i[:3]
        if len(i) > 4 and i[4] > 1:                                     # clips in lock step: clip by clip with the pointers advanced
            K, ncell = P - 1, (h // 4) * (w // 4)
            for c in range(i[4]):
                pc = list(p)
                pc[0], pc[1] = p[0] + 4 * c * K * h * w, p[1] + 4 * c * P * 16 * h * w
                pc[2] = p[2] + 4 * c * P * 16 * h * w if p[2] else 0
                if flags & 4:
                    pc[3], pc[4] = p[3] + 4 * c * K * ncell, p[4] + 2 * c * K * ncell * i[3]
                self._op_11(flags, i[:4] + [1] + i[5:], f, pc)
            return
        if flags & 1:                                                   # SEG_AGG fused: p0 = raw logits [P-1,h,w]
            pr = torch.sigmoid(view(p[0], F32, (P - 1, h * w)))
            agg = torch.cat([_clamp_logit(torch.prod(1 - pr, dim=0, keepdim=True)), _clamp_logit(pr)], 0).view(P, h, w)
        else:
            agg = view(p[0], F32, (P, h, w))
        up = _bilinear(agg, 0.25)
        if p[2]:
            view(p[2], F32, (P, 4 * h, 4 * w)).copy_(up)
        prob = view(p[1], F32, (P, 4 * h, 4 * w))
        prob.copy_(torch.softmax(up, dim=0))
        if flags & 4:                                                   # + MASK_DOWN of prob[1:] (r = 16)
            K, H, W = P - 1, 4 * h, 4 * w
            m16 = F.avg_pool2d(prob[1:].unsqueeze(0), 16)[0]
            view(p[3], F32, (K, H // 16, W // 16)).copy_(m16)
            ld = i[3]
            out = view(p[4], BF16, (K, H // 16, W // 16, 2), ((H // 16) * (W // 16) * ld, (W // 16) * ld, ld, 1))
            out[..., 0] = m16
            out[..., 1] = (m16.sum(0, keepdim=True) - m16).clamp(0, 1)

    # ---- MASK_MERGE / AGG_SOFTMAX ----------------------------------------------------------------
    def _op_12(self, flags, i, f, p):
        h0, w0, H, W, pl, pt, Knew, Kold, nfloat = i[:9]
        src = view(p[2], I32, (Knew,))
        planes = view(p[3], F32, (Knew, H, W))
        pred = view(p[1], F32, (Kold + 1, H, W)) if p[1] else None
        if flags & 1:
            fm = torch.zeros(nfloat, H, W)
            fm[:, pt:pt + h0, pl:pl + w0] = view(p[0], F32, (nfloat, h0, w0))
            covered = fm.max(0)[0] > 0.5
        else:
            idx = torch.zeros(H, W, dtype=torch.int32)
            idx[pt:pt + h0, pl:pl + w0] = view(p[0], I32, (h0, w0))
            covered = idx > 0
        for t in range(Knew):
            s = int(src[t])
            if s >= 0:
                planes[t] = fm[s] if flags & 1 else (idx == s).float()
            elif pred is not None and t < Kold:
                planes[t] = torch.where(covered, torch.zeros(()), pred[t + 1])
            else:
                planes[t] = 0

    def _op_13(self, flags, i, f, p):
        K, HW = i[:2]
        pl = view(p[0], F32, (K, HW))
        lg = torch.cat([_clamp_logit(torch.prod(1 - pl, dim=0, keepdim=True)), _clamp_logit(pl)], 0)
        view(p[1], F32, (K + 1, HW)).copy_(torch.softmax(lg, dim=0))

    # ---- LINEAR / LAYERNORM / QUERY_INIT ------------------------------------------------------------
    def _op_14(self, flags, i, f, p):
        M, N, Kd, ldx, ldy, add_rows, add_cols = i[:7]
        x = view(p[0], F32, (M, Kd), (ldx, 1)).clone()
        if flags & 2:
            x = F.layer_norm(x, (Kd,), view(p[6], F32, (Kd,)), view(p[7], F32, (Kd,)), f[0] if f[0] > 0 else 1e-5)
            if p[8]:
                view(p[8], F32, (M, Kd)).copy_(x)
        w = view(p[2], BF16, (N, Kd)).float()
        y = x @ w.t()
        if p[1]:
            R = max(add_rows, 1)
            xa = view(p[1], F32, (R, Kd))
            ya = (x + xa[torch.arange(M) % R]) @ w.t()
            if add_cols > 0:
                y[:, :add_cols] = ya[:, :add_cols]
            else:
                y = ya
        if p[3]:
            y = y + view(p[3], F32, (N,))
        if flags & 1:
            y = F.relu(y)
        if p[4]:
            y = y + view(p[4], F32, (M, N))
        view(p[5], F32, (M, N), (ldy, 1)).copy_(y)

    def _op_15(self, flags, i, f, p):
        M, C = i[:2]
        x = view(p[0], F32, (M, C)).clone()
        view(p[3], F32, (M, C)).copy_(F.layer_norm(x, (C,), view(p[1], F32, (C,)), view(p[2], F32, (C,)), 1e-5))

    def _op_16(self, flags, i, f, p):
        rows, C = i[:2]
        om = view(p[0], F32, (rows, C + 1))
        vals = om[:, :C] / (om[:, C:] + 1e-4)
        if flags & 1:                                                   # with the two linears (+ side job: clear p9, i2 x 16 bytes)
            if p[9] and i[2] > 0:
                view(p[9], I64, (2 * i[2],)).zero_()
            for y, w, b, r in ((1, 3, 4, 5), (2, 6, 7, 8)):
                out = vals @ view(p[w], BF16, (C, C)).float().t()
                if p[b]:
                    out = out + view(p[b], F32, (C,))
                if p[r]:
                    out = out + view(p[r], F32, (rows, C))
                view(p[y], F32, (rows, C)).copy_(out)
            return
        view(p[1], F32, (rows, C)).copy_(vals)

    QACC = 4294967296.0

    @classmethod
    def _partial_sum(cls, x, flags, i, p, M, always=False):
        """flags&4 (QFFN: whenever p10 is given): x_eff = x + p11 + p10 / 2^32 (p10: int64 fixed-point accumulator [M, 256])."""
        if (flags & 4 or always) and p[10]:
            if p[11]:
                x = x + view(p[11], F32, (256,))
            x = x + (view(p[10], I64, (M, 256)).double() / cls.QACC).float()
        return x

    @classmethod
    def _acc_add(cls, ptr, M, val):
        acc = view(ptr, I64, (M, 256))
        acc += torch.round(val.double() * cls.QACC).to(torch.int64)

    @classmethod
    def _out_parts(cls, flags, p, att, M, heads):
        """flags&8: p13 (int64 [M, 256]) += att . Wo^T in fixed point, head by head (the bias is the consumer's business)."""
        if flags & 8:
            Wo = view(p[12], BF16, (256, 256)).float()
            a = att.reshape(M, 256)
            for h in range(heads):
                cls._acc_add(p[13], M, a[:, 32 * h:32 * h + 32] @ Wo[:, 32 * h:32 * h + 32].t())

    def _proj_rows(self, p, M, ldx, ln_out_slot, flags=0, i=None):
        """Fused projection operands (ops.OpList._proj): -> (LN(x) + emb, LN(x), W [N,256] fp32, bias)."""
        x = view(p[0 if ln_out_slot != -1 else 1], F32, (M, 256), (ldx, 1)).clone()
        x = self._partial_sum(x, flags, i, p, M)
        if p[8]:
            x = F.layer_norm(x, (256,), view(p[8], F32, (256,)), view(p[9], F32, (256,)), 1e-5)
            if ln_out_slot >= 0 and p[ln_out_slot]:
                view(p[ln_out_slot], F32, (M, 256)).copy_(x)
        xa = x + view(p[7], F32, (M, 256)) if p[7] else x
        return xa, x

    # ---- AUX_MASK / attention ----------------------------------------------------------------------------
    def _op_17(self, flags, i, f, p):
        K, HW = i[:2]
        pr = torch.sigmoid(view(p[0], F32, (K, HW)))
        lg = torch.cat([_clamp_logit(torch.prod(1 - pr, dim=0, keepdim=True)), _clamp_logit(pr)], 0)
        fg = lg[1:] >= lg.max(0, keepdim=True)[0]
        view(p[1], U8, (K, HW)).copy_(fg.to(torch.uint8))
        nfg = view(p[2], I32, (K,))
        nfg += fg.sum(1).to(torch.int32)

    def _op_18(self, flags, i, f, p):
        K, Q, HW, C, heads, ldkv, voff = i[:7]
        hd = C // heads
        if flags & 16:                                                  # chain form: q given, projected and scaled by 1 / sqrt(hd)
            q = (view(p[0], F32, (K, Q, C)) * math.sqrt(hd)).view(K, Q, heads, hd).transpose(1, 2)
        elif flags & 2:                                                 # q projection fused
            xa, _ = self._proj_rows(p, K * Q, i[7] or 256, 3, flags, i)
            qp = xa @ view(p[5], BF16, (C, 256)).float().t() + (view(p[6], F32, (C,)) if p[6] else 0)
            q = qp.view(K, Q, heads, hd).transpose(1, 2)
        else:
            q = view(p[0], F32, (K, Q, C)).view(K, Q, heads, hd).transpose(1, 2)
        hs = (i[8] if len(i) > 8 and i[8] > 0 and (flags & 8) else hd)          # chain form: elements between the heads inside a pixel row
        k = view(p[1], BF16, (K, HW, heads, hd), (HW * ldkv, ldkv, hs, 1)).float().transpose(1, 2)
        v = view(p[1] + 2 * voff, BF16, (K, HW, heads, hd), (HW * ldkv, ldkv, hs, 1)).float().transpose(1, 2)
        if flags & 1:                                                   # AUX_MASK fused: p2 = logits
            kg = i[9] if (len(i) > 9 and i[9] > 0 and (flags & 8)) else K   # chain form: objects per clip (clips in lock step)
            pr = torch.sigmoid(view(p[2], F32, (K // kg, kg, HW)))
            lg = torch.cat([_clamp_logit(torch.prod(1 - pr, dim=1, keepdim=True)), _clamp_logit(pr)], 1)
            fg = (lg[:, 1:] >= lg.max(1, keepdim=True)[0]).reshape(K, HW)
            nfg = fg.sum(1).to(torch.int32)
        else:
            fg = view(p[2], U8, (K, HW)).bool()
            nfg = view(p[3], I32, (K,))
        att = (q @ k.transpose(-1, -2)) / math.sqrt(hd)                 # [K,heads,Q,HW]
        for kk in range(K):
            n = int(nfg[kk])
            if n != 0:
                att[kk, :, :Q // 2, :] = att[kk, :, :Q // 2, :].masked_fill(~fg[kk], float('-inf'))
            if n != HW:
                att[kk, :, Q // 2:, :] = att[kk, :, Q // 2:, :].masked_fill(fg[kk], float('-inf'))
        out = (att.softmax(-1) @ v).transpose(1, 2).reshape(K, Q, C)
        if p[4]:
            view(p[4], F32, (K, Q, C)).copy_(out)
        self._out_parts(flags, p, out, K * Q, heads)

    def _op_19(self, flags, i, f, p):
        K, Q, C, heads, ldqk, ldv = i[:6]
        hd = C // heads
        if flags & 2:                                                   # qkv projection fused
            xa, xp = self._proj_rows(p, K * Q, i[6] or 256, 3, flags, i)
            Wt = view(p[5], BF16, (3 * C, 256)).float()
            b = view(p[6], F32, (3 * C,)) if p[6] else torch.zeros(3 * C)
            q = (xa @ Wt[:C].t() + b[:C]).view(K, Q, heads, hd).transpose(1, 2)
            k = (xa @ Wt[C:2 * C].t() + b[C:2 * C]).view(K, Q, heads, hd).transpose(1, 2)
            v = (xp @ Wt[2 * C:].t() + b[2 * C:]).view(K, Q, heads, hd).transpose(1, 2)
        else:
            ldqk, ldv = ldqk or 2 * C, ldv or C
            qk = view(p[0], F32, (K, Q, 2 * C), (Q * ldqk, ldqk, 1))
            q = qk[..., :C].reshape(K, Q, heads, hd).transpose(1, 2)
            k = qk[..., C:].reshape(K, Q, heads, hd).transpose(1, 2)
            v = view(p[1], F32, (K, Q, C), (Q * ldv, ldv, 1)).reshape(K, Q, heads, hd).transpose(1, 2)
        att = ((q @ k.transpose(-1, -2)) / math.sqrt(hd)).softmax(-1)
        out = (att @ v).transpose(1, 2).reshape(K, Q, C)
        if p[2]:
            view(p[2], F32, (K, Q, C)).copy_(out)
        self._out_parts(flags, p, out, K * Q, heads)

    def _op_20(self, flags, i, f, p):
        K, Q, HW, C, heads, ldq, ldkv = i[:7]
        hd = C // heads
        ldkv = ldkv or C
        q = view(p[0], BF16, (K, HW, C), (HW * ldq, ldq, 1)).float().view(K, HW, heads, hd).transpose(1, 2)
        if flags & 2:                                                   # kv projection fused (no LayerNorm; p1 = x)
            x = self._partial_sum(view(p[1], F32, (K * Q, 256), (i[7] or 256, 1)), flags, i, p, K * Q)
            xa = x + view(p[7], F32, (K * Q, 256)) if p[7] else x
            if flags & 16:                                              # the next block's ATTN_Q2P queries from the same rows
                xn = F.layer_norm(x, (256,), view(p[8], F32, (256,)), view(p[9], F32, (256,)), 1e-5)
                view(p[15], F32, (K * Q, 256)).copy_(xn)
                qn = (xn + view(p[7], F32, (K * Q, 256)) if p[7] else xn) @ view(p[12], BF16, (256, 256)).float().t()
                if p[13]:
                    qn = qn + view(p[13], F32, (256,))
                view(p[14], F32, (K * Q, 256)).copy_(qn / math.sqrt(32))
            Wt = view(p[5], BF16, (2 * C, 256)).float()
            b = view(p[6], F32, (2 * C,)) if p[6] else torch.zeros(2 * C)
            k = (xa @ Wt[:C].t() + b[:C]).view(K, Q, heads, hd).transpose(1, 2)
            v = (x @ Wt[C:].t() + b[C:]).view(K, Q, heads, hd).transpose(1, 2)
        else:
            k = view(p[1], F32, (K, Q, C), (Q * ldkv, ldkv, 1)).reshape(K, Q, heads, hd).transpose(1, 2)
            v = view(p[2], F32, (K, Q, C), (Q * ldkv, ldkv, 1)).reshape(K, Q, heads, hd).transpose(1, 2)
        att = ((q @ k.transpose(-1, -2)) / math.sqrt(hd)).softmax(-1)
        o = (att @ v).transpose(1, 2).reshape(K, HW, C)
        if flags & 32:                                                  # + the 1x1 conv behind it and the residual: p2 = [Wo bf16 | bias f32], p4 = residual
            Wo = view(p[2], BF16, (C, C)).float()
            bo = view(p[2] + 2 * C * C, F32, (C,))
            o = o.to(torch.bfloat16).float() @ Wo.t() + bo + view(p[4], BF16, (K, HW, C)).float()
        view(p[3], BF16, (K, HW, C)).copy_(o)

    # ---- SUMMARIZE / ADD_PE ---------------------------------------------------------------------------------
    def _op_21(self, flags, i, f, p):
        K, HW, C, Q = i[:4]
        ldf, ldw = (i[4] or C), (i[5] or Q)
        feat = view(p[0], F32 if flags & 1 else BF16, (K, HW, C), (HW * ldf, ldf, 1)).float()
        wl = view(p[1], F32, (K, HW, Q), (HW * ldw, ldw, 1))
        m = view(p[2], F32, (K, HW)).unsqueeze(-1)
        rep = torch.cat([m.expand(-1, -1, Q // 2), (1 - m).expand(-1, -1, Q // 2)], -1)
        wgt = torch.sigmoid(wl) * rep
        out = view(p[3], F32, (K, Q, C + 1))
        out[..., :C] = torch.einsum('kpq,kpc->kqc', wgt, feat)
        out[..., C] = wgt.sum(1)

    def _op_22(self, flags, i, f, p):
        B, n = i[:2]
        x = view(p[0], BF16, (B, n)).float()
        view(p[2], BF16, (B, n)).copy_(x + view(p[1], BF16, (n,)).float())

    # ---- affinity -----------------------------------------------------------------------------------------------
    def _op_23(self, flags, i, f, p):
        n = i[0]
        key = view(p[0], F32, (n, 64))
        if flags & 1:
            e = view(p[1], F32, (n, 64))
            v = torch.cat([-e, 2 * key * e], 1)
            view(p[4], F32, (n,)).copy_((e * key * key).sum(1))
        else:
            v = torch.cat([key * key, key], 1)
            view(p[4], F32, (n,)).copy_(view(p[1], F32, (n,)) * 0.125)
        hi = v.to(torch.bfloat16)
        lo = (v - hi.float()).to(torch.bfloat16)
        view(p[2], BF16, (n, 128)).copy_(hi)
        view(p[3], BF16, (n, 128)).copy_(lo)

    @staticmethod
    def _ranges(i):
        nr = i[2]
        return [(i[3 + 2 * r], i[4 + 2 * r]) for r in range(nr)]

    def _scores(self, i, p):
        """[(slot indices, S[n_r, HW])] per range, computed like the kernel (3-term split bf16)."""
        HW = i[0]
        Bh = view(p[3], BF16, (HW, 128)).float()
        Bl = view(p[4], BF16, (HW, 128)).float()
        c = view(p[5], F32, (HW,))
        out = []
        for (s, n) in self._ranges(i):
            Ah = view(p[0] + 2 * 128 * s, BF16, (n, 128)).float()
            Al = view(p[1] + 2 * 128 * s, BF16, (n, 128)).float()
            sc = view(p[2] + 4 * s, F32, (n,))
            acc = (Ah @ Bl.t() + Al @ Bh.t()) + Ah @ Bh.t()
            out.append((torch.arange(s, s + n), sc[:, None] * (acc - c[None, :])))
        return out

    def _op_24(self, flags, i, f, p):
        """AFF_SCORE.  i[16] > 0: i[1] / i[16] frames of i[16] query rows each (i[0] real ones), every per-query array indexed by the
        stacked row -- interpreted frame by frame with the pointers advanced."""
        HW, HWp = i[0], i[1]
        G, cap, mode = i[9], i[10], i[11]
        HWpf = i[16] if i[16] > 0 else HWp
        Gld = -(-G // 64) * 64
        for fr in range(HWp // HWpf):
            r0 = fr * HWpf
            pf = list(p)
            pf[3], pf[4], pf[5] = p[3] + 256 * r0, p[4] + 256 * r0, p[5] + 4 * r0
            if flags & 4:                                               # (ABI 4) frame fr reads bank fr % i[17]: (A_hi, A_lo, scale) from the table p[11]
                assert i[17] >= 2 and (HWp // HWpf) % i[17] == 0 and HWpf % 128 == 0 and i[12] == 2
                pf[0], pf[1], pf[2] = [int(v) for v in view(p[11], U64, (i[17], 3))[fr % i[17]]]
            sc = self._scores(i, pf)
            if mode == 0:
                gmax = view(p[6] + 4 * r0 * Gld, F32, (HWpf, Gld))
                g = 0
                for (_, S) in sc:
                    n = S.shape[0]
                    T = -(-n // 16)
                    Sp = torch.full((T * 16, HW), float('-inf'))
                    Sp[:n] = S
                    gmax[:HW, g:g + T] = Sp.view(T, 16, HW).max(1)[0].t()
                    g += T
            else:
                tau = view(p[6] + 4 * r0, F32, (HW,))
                thr = tau - tau.abs() * 1e-6 - 1e-30
                thr = torch.where(torch.isinf(tau), torch.full_like(tau, float('-inf')), thr)
                cv = view(p[7] + 4 * r0 * cap, F32, (HW, cap))
                ci = view(p[8] + 4 * r0 * cap, I32, (HW, cap))
                cnt = view(p[9] + 4 * 32 * r0, I32, (HW, 32))[:, 0]
                for (slots, S) in sc:
                    for j in range(HW):
                        sel = torch.nonzero(S[:, j] >= thr[j]).flatten()
                        c0 = int(cnt[j])
                        m = min(len(sel), max(cap - c0, 0))
                        cv[j, c0:c0 + m] = S[sel[:m], j]
                        ci[j, c0:c0 + m] = slots[sel[:m]].to(torch.int32)
                        cnt[j] = c0 + len(sel)

    def _op_25(self, flags, i, f, p):
        HW, HWp, G, k = i[:4]
        frames = i[6] if i[6] > 1 else 1
        Gld = -(-G // 64) * 64
        if p[2]:                                                        # side jobs: clear pass 1's counters, advance life counters
            view(p[2], I32, (HW if frames == 1 else frames * HWp, 32))[:, 0] = 0
        for slot, n in ((3, i[4]), (4, i[5])):
            if p[slot] and n > 0:
                if slot == 3 and (flags & 1):
                    view(p[slot], F32, (n,)).zero_()                     # the usage side buffer of a look-ahead read-out
                else:
                    view(p[slot], F32, (n,)).add_(1.0)
        for fr in range(frames):
            r0 = fr * HWp
            gmax = view(p[0] + 4 * r0 * Gld, F32, (HWp, Gld))[:HW, :G]
            tau = view(p[1] + 4 * r0, F32, (HW,))
            if G < k:
                tau.fill_(float('-inf'))
            else:
                tau.copy_(gmax.topk(k, dim=1)[0][:, -1])

    def _op_26(self, flags, i, f, p):
        HW, cap, topk, K, CV = i[:5]
        frames = i[5] if i[5] > 1 else 1
        HWpf, ustride = (i[6], i[7]) if frames > 1 else (HW, 0)
        nb = i[8] if i[8] > 1 else 1
        assert frames % nb == 0
        vptrs_all = view(p[3], U64, (nb, K))
        for fr in range(frames):
            vptrs = vptrs_all[fr % nb]
            r0 = fr * HWpf
            cv = view(p[0] + 4 * r0 * cap, F32, (HW, cap))
            ci = view(p[1] + 4 * r0 * cap, I32, (HW, cap))
            cnt = view(p[2] + 4 * 32 * r0, I32, (HW, 32))[:, 0]
            ufx = bool(flags & 1)                                          # usage counters: unsigned 64-bit fixed point (2^-40)
            usage = p[4] + (8 if ufx else 4) * fr * ustride if p[4] else 0
            out = view(p[5] + 2 * fr * K * HW * CV, BF16, (K, HW, CV))
            for j in range(HW):
                n = min(int(cnt[j]), cap)
                if int(cnt[j]) > cap:
                    view(p[6], I32, (1,))[0] += 1
                v, idx = cv[j, :n], ci[j, :n].long()
                # descending by value, ties -> lower slot
                order = sorted(range(n), key=lambda t: (-float(v[t]), int(idx[t])))[:topk]
                order = torch.tensor(order, dtype=torch.long)
                sv, si = v[order], idx[order]
                w = torch.exp(sv - sv[0])
                w = w / w.sum()
                if usage and ufx:
                    u = view(usage, U64, (int(si.max()) + 1,))
                    u.index_add_(0, si, (w.float() * 1099511627776.0).to(torch.int64))       # (the kernel truncates the f32 product)
                elif usage:
                    u = view(usage, F32, (int(si.max()) + 1,))
                    u.index_add_(0, si, w)
                for o in range(K):
                    V = view(int(vptrs[o]), BF16, (int(si.max()) + 1, CV)).float()
                    out[o, j] = (w[:, None] * V[si]).sum(0).to(torch.bfloat16)

    # ---- misc --------------------------------------------------------------------------------------------------------
    def _op_27(self, flags, i, f, p):
        view(p[0], I32, (i[0],)).fill_(i[1])

    def _op_28(self, flags, i, f, p):
        rows, rb, ss, ds = i[:4]
        src = view(p[0], U8, (rows, rb), (ss, 1)).clone()
        view(p[1], U8, (rows, rb), (ds, 1)).copy_(src)

    def _op_42(self, flags, i, f, p):                    # BANK_WRITE
        for k in range(6):
            if p[2 * k] and p[2 * k + 1] and i[k] > 0:
                view(p[2 * k + 1], I32, (i[k],)).copy_(view(p[2 * k], I32, (i[k],)).clone())
        for k in range(2):
            if p[12 + k] and i[6 + k] > 0:
                view(p[12 + k], I32, (i[6 + k],)).fill_(i[8 + k])

    def _op_29(self, flags, i, f, p):
        n = i[0]
        y = view(p[1], F32, (n,))
        y += f[0] * view(p[0], F32, (n,))

    def _op_30(self, flags, i, f, p):
        if p[0] and i[0] > 0:
            view(p[0], F32, (i[0],)).add_(1.0)
        if p[1] and i[1] > 0:
            view(p[1], F32, (i[1],)).add_(1.0)
        if p[2] and p[3] and i[2] > 0 and (flags & 1):                   # delta: unsigned 64-bit fixed point (2^-40), one rounding
            view(p[2], F32, (i[2],)).add_((view(p[3], U64, (i[2],)).double() * 2.0 ** -40).float())
            if flags & 2:
                view(p[3], U64, (i[2],)).zero_()
        elif p[2] and p[3] and i[2] > 0:
            view(p[2], F32, (i[2],)).add_(view(p[3], F32, (i[2],)))

    def _op_31(self, flags, i, f, p):
        n, k = i[:2]
        u = view(p[0], F32, (n,)) / view(p[1], F32, (n,))
        order = sorted(range(n), key=lambda t: (-float(u[t]), t))[:k]
        order_t = torch.tensor(order, dtype=torch.int32)
        view(p[2], I32, (k,)).copy_(order_t)
        for slot, words in ((4, i[2]), (6, i[3])):                       # side jobs: row gathers through the new order, a cleared buffer
            if p[slot] and words > 0:
                src = view(p[slot], I32, (n, words))
                view(p[slot + 1], I32, (k, words)).copy_(src[order_t.long()].clone())
        if p[8] and i[4] > 0:
            view(p[8], I32, (i[4],)).zero_()

    def _op_32(self, flags, i, f, p):
        k, rb, ss, ds = i[:4]
        order = view(p[1], I32, (k,)).long()
        nsrc = int(order.max()) + 1
        src = view(p[0], U8, (nsrc, rb), (ss, 1))
        view(p[2], U8, (k, rb), (ds, 1)).copy_(src[order])

    def _op_33(self, flags, i, f, p):
        """CONSOL_AFF: S[p, i] = similarity of candidate i and prototype p, -inf in the padding up to ldS (the column maxima p5 are an
        internal hand-over to CONSOL_READ: the interpreter's CONSOL_READ takes its maxima from S)."""
        n, P, ldS = i[:3]
        ck, cs = view(p[0], F32, (n, 64)), view(p[1], F32, (n,))
        pk, pe = view(p[2], F32, (P, 64)), view(p[3], F32, (P, 64))
        a_sq = (ck * ck) @ pe.t()
        two_ab = 2 * (ck @ (pk * pe).t())
        b_sq = (pe * pk * pk).sum(1)[None, :]
        sim = (-a_sq + two_ab - b_sq) * cs[:, None] * 0.125           # [n,P]
        S = view(p[4], F32, (P, ldS))
        S.fill_(float('-inf'))
        S[:, :n] = sim.t()

    def _op_34(self, flags, i, f, p):
        """CONSOL_READ: softmax over the candidates, applied to every object's values and to the shrinkage."""
        n, P, C, K, ldS, src, dst = i[:7]
        aff = torch.softmax(view(p[0], F32, (P, ldS))[:, :n], dim=1)
        vptrs = view(p[2], U64, (K,))
        for o in range(K):
            bank = view(int(vptrs[o]), BF16, (max(src + n, dst + P), C))
            proto = (aff @ bank[src:src + n].float()).to(torch.bfloat16)
            bank[dst:dst + P] = proto
        if p[5]:
            view(p[5], F32, (P,)).copy_(aff @ view(p[3], F32, (n,)))

    def _op_41(self, flags, i, f, p):                                   # STEM = IMG_PREP + conv 7x7 s2 p3 (+ bias) + maxpool 3x3 s2 p1 (+ relu)
        h0, w0, H, W, pl, pt, K, Kpad = i[:8]
        K = K if p[1] else 1
        if i[8] > 1:                                                    # several frames per launch: frame by frame with the pointers advanced
            for fr in range(i[8]):
                pf = list(p[:5]) + [0] * 11
                pf[0] = p[0] if fr == 0 else p[4 + fr]
                pf[1] = p[1] + 4 * fr * i[9] if p[1] else 0
                pf[4] = p[4] + 2 * fr * K * (H // 4) * (W // 4) * 64
                self._op_41(flags, i[:8] + [0] * 16, f, pf)
            return
        img = view(p[0], F32, (3, h0, w0))
        full = torch.zeros(3, H, W)
        full[:, pt:pt + h0, pl:pl + w0] = img
        full = (full - torch.tensor(f[0:3]).view(3, 1, 1)) / torch.tensor(f[3:6]).view(3, 1, 1)
        x = torch.zeros(K, 8, H, W)
        x[:, 0:3] = full
        if p[1]:
            m = view(p[1], F32, (K, H, W))
            x[:, 3] = m
            x[:, 4] = (m.sum(0, keepdim=True) - m).clamp(0, 1)
        x = x.to(torch.bfloat16).float()
        wgt = unpack_conv_weight(view(p[2], BF16, (64, Kpad)), 64, 7, 7, 8)
        y = F.conv2d(x, wgt, view(p[3], F32, (64,)), stride=2, padding=3).to(torch.bfloat16).float()
        y = F.max_pool2d(y, 3, 2, 1)
        if flags & 1:
            y = F.relu(y)
        view(p[4], BF16, (K, H // 4, W // 4, 64)).copy_(y.permute(0, 2, 3, 1))

    def _op_40(self, flags, i, f, p):                                   # QFFN
        rows, FF = i[:2]
        x = self._partial_sum(view(p[0], F32, (rows, 256)).clone(), flags, i, p, rows, always=True)
        if p[1]:
            view(p[1], F32, (rows, 256)).copy_(x)
        xn = F.layer_norm(x, (256,), view(p[2], F32, (256,)), view(p[3], F32, (256,)), 1e-5)
        W1 = view(p[4], BF16, (FF, 256)).float()
        b1 = view(p[5], F32, (FF,)) if p[5] else torch.zeros(FF)
        W2 = view(p[6], BF16, (256, FF)).float()
        hid = torch.relu(xn @ W1.t() + b1)
        HS = i[2] or 64
        for s_ in range(FF // HS):
            self._acc_add(p[7], rows, hid[:, HS * s_:HS * s_ + HS] @ W2[:, HS * s_:HS * s_ + HS].t())

    def _op_38(self, flags, i, f, p):
        rows, W, slds, dlds = i[:4]
        src = view(p[0], F32, (rows, W), (slds, 1)).clone()
        dst = view(p[1], F32, (rows, W), (dlds, 1))
        out = f[0] * torch.flip(src, dims=[-1])
        dst.copy_(out + f[1] * dst if f[1] != 0 else out)

    def _op_37(self, flags, i, f, p):
        C, H, W, OH, OW, plane, ldrow = i[:7]
        src = view(p[0], F32, (C, H, W), (plane, ldrow, 1)).clone().unsqueeze(0)
        if flags & 1:
            out = F.interpolate(src, size=(OH, OW), mode='nearest-exact')
        else:
            out = F.interpolate(src, size=(OH, OW), mode='bilinear', align_corners=False)
        view(p[1], F32, (C, OH, OW)).copy_(out[0])

    def _op_36(self, flags, i, f, p):
        P, H, W, plane, ldrow = i[:5]
        prob = view(p[0], F32, (P, H, W), (plane, ldrow, 1))
        lut = view(p[1], I32, (P,)).long()
        ids = lut[prob.argmax(0)]
        dt = {0: U8, 1: I32, 2: I64}[flags & 3]
        view(p[2], dt, (H, W)).copy_(ids)

    def _op_35(self, flags, i, f, p):
        n = i[0]
        if flags & 1:
            view(p[1], F32, (n,)).copy_(view(p[0], BF16, (n,)).float())
        else:
            view(p[1], BF16, (n,)).copy_(view(p[0], F32, (n,)))
