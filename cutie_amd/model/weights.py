"""Weight preparation for the HIP kernels: BatchNorm folding, NHWC/K-major packing, bf16 conversion.

Runs once at model load (host side, torch used for the one-off reshuffle), never on the per-frame path.
Reference semantics folded here: eval-mode BatchNorm2d (PixelEncoder/MaskEncoder force eval,
big_modules.py:56-61,184-189), Conv2d/Linear weight layouts, nn.MultiheadAttention packed in_proj.
"""
import torch

BN_EPS = 1e-5


class PackedConv:
    """bf16 weight [CoutPad, Kpad] with k = (kh*KW + kw)*CinPadded + c; f32 bias [Cout] or None."""
    __slots__ = ('weight', 'bias', 'cout', 'cin_padded', 'kh', 'kw', 'kpad', 'cin_real')

    def __init__(self, weight, bias, cout, cin_padded, kh, kw, kpad, cin_real=None):
        self.weight, self.bias, self.cout, self.cin_padded = weight, bias, cout, cin_padded
        self.kh, self.kw, self.kpad = kh, kw, kpad
        self.cin_real = cin_padded if cin_real is None else cin_real      # un-padded channels (flop accounting)


class PackedLinear:
    """bf16 weight [N, Kd] (torch Linear layout), f32 bias [N] or None."""
    __slots__ = ('weight', 'bias', 'n', 'kd')

    def __init__(self, weight, bias, n, kd):
        self.weight, self.bias, self.n, self.kd = weight, bias, n, kd


def fold_bn(w, sd, bn):
    g, b = sd[bn + '.weight'].float(), sd[bn + '.bias'].float()
    m, v = sd[bn + '.running_mean'].float(), sd[bn + '.running_var'].float()
    s = g / torch.sqrt(v + BN_EPS)
    return w * s.view(-1, 1, 1, 1), b - m * s


def pack_conv(w, bias, device, segs=None):
    """w fp32 [Cout, Cin, KH, KW]; segs = [(real, padded), ...] describes how the input channels are laid out
    in the (virtually concatenated) NHWC sources, each segment zero-padded to a multiple of 8."""
    w = w.float()
    cout, cin, kh, kw = w.shape
    if segs is None:
        segs = [(cin, -(-cin // 8) * 8)]
    assert sum(r for r, _ in segs) == cin
    parts, c0 = [], 0
    for real, padded in segs:
        assert padded % 8 == 0 and padded >= real
        part = w[:, c0:c0 + real]
        if padded > real:
            part = torch.cat([part, torch.zeros(cout, padded - real, kh, kw)], 1)
        parts.append(part)
        c0 += real
    w = torch.cat(parts, 1)
    cin_p = w.shape[1]
    k = kh * kw * cin_p
    kpad = -(-k // 128) * 128                  # multiple of every BK the kernel family uses (32/64/128)
    coutpad = -(-cout // 128) * 128
    flat = w.permute(0, 2, 3, 1).reshape(cout, k)
    packed = torch.zeros(coutpad, kpad)
    packed[:cout, :k] = flat
    return PackedConv(packed.to(torch.bfloat16).to(device).contiguous(),
                      None if bias is None else bias.float().to(device).contiguous(), cout, cin_p, kh, kw, kpad, cin_real=cin)


def pack_linear(w, bias, device):
    w = w.float()
    n, kd = w.shape
    assert kd % 8 == 0
    return PackedLinear(w.to(torch.bfloat16).to(device).contiguous(),
                        None if bias is None else bias.float().to(device).contiguous(), n, kd)


def linear_as_conv(w, bias, device):
    """nn.Linear applied to pixel tokens == 1x1 conv."""
    return pack_conv(w.float().view(w.shape[0], w.shape[1], 1, 1), bias, device)


def out_proj_blob(w):
    """[Wo bf16 [256, 256] (out, in) | bias f32 [256]] as ONE byte tensor: the operand of ATTN_P2Q's fused output projection (flags&32;
    the op has two free pointer slots, the residual takes the other).  w: the PackedConv of the 1x1 conv it replaces."""
    assert w.cout == 256 and w.kh == 1 and w.kw == 1 and w.cin_padded == 256 and w.bias is not None
    return torch.cat([w.weight[:256, :256].contiguous().view(torch.uint8).reshape(-1), w.bias.contiguous().view(torch.uint8).reshape(-1)]).contiguous()
