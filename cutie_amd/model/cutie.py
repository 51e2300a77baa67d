"""``CUTIE`` -- drop-in mirror of the reference ``cutie.model.cutie.CUTIE`` (cutie/model/cutie.py:18-260)
whose compute runs as HIP launch plans on MI355X.

Same constructor, same six public methods, same ``state_dict`` keys (527 tensors) so reference checkpoints
load unchanged through ``load_weights``.  Differences a caller can observe:
  * tensors returned by the facade methods keep the reference's *logical* shapes (``[1,C,H,W]``,
    ``[1,K,C,h,w]``) but are bf16 views of NHWC device buffers (like autocast would return half tensors);
    keys/shrinkage/selection, logits, probabilities, summaries and the sensory state stay fp32.
  * the sensory state passed to ``segment`` / ``encode_mask`` is updated in place (and returned).
  * inference only: ``read_memory`` / ``compute_aux`` / ``forward`` (training paths) raise.
There is no CPU fallback: every method needs libcutie_hip.so and a HIP device.
"""
import logging
from typing import Dict, Iterable, List

import torch
import torch.nn as nn

from .. import _lib, frame_context
from . import plans
from .param_spec import build_spec
from .weights import fold_bn, pack_conv, pack_linear, linear_as_conv, out_proj_blob

log = logging.getLogger()
BF16, F32 = torch.bfloat16, torch.float32


# ---- logical <-> physical layout helpers -----------------------------------------------------------
def nhwc_of(t, C=None):
    """logical [B,C,H,W] -> physical contiguous bf16 [B,H,W,C] (no copy when it already is our layout)."""
    p = t.permute(0, 2, 3, 1)
    if p.dtype != BF16 or not p.is_contiguous():
        p = p.to(BF16).contiguous()
    return p


def group_nhwc_of(t, dtype=BF16):
    """logical [1,K,C,h,w] -> physical [K,h,w,C]."""
    assert t.shape[0] == 1, 'batch size 1 only (flip_aug is not supported yet)'
    p = t[0].permute(0, 2, 3, 1)
    if p.dtype != dtype or not p.is_contiguous():
        p = p.to(dtype).contiguous()
    return p


def f32c(t):
    """t as contiguous fp32 -- without a torch call when it already is (host time: a no-op .to() costs ~1 us, five per frame)."""
    if t.dtype != F32:
        t = t.to(F32)
    return t if t.is_contiguous() else t.contiguous()


def logical(p):
    """physical [B,H,W,C] -> logical [B,C,H,W] view."""
    return p.permute(0, 3, 1, 2)


def group_logical(p):
    """physical [K,h,w,C] -> logical [1,K,C,h,w] view."""
    return p.permute(0, 3, 1, 2).unsqueeze(0)


class Engine:
    """Device-resident packed weights + plan cache for one CUTIE module state."""

    def __init__(self, sd: Dict[str, torch.Tensor], m, device):
        self.m, self.device = m, device
        self.devstr = str(device)                   # (pool keys; str(device) per call is host time)
        self.one_lane = False                       # the look-ahead lanes on the caller's own stream (parallel.run_concurrent sets it)
        self.w = {}
        self._plans = {}
        self.pool = plans.SlotPool()                # per-frame plan outputs at stable addresses (HIP-graph replay)
        self.enc_specs = {}                         # (h, w) -> (pool key, output specs) of the image encoder: built once (host time)
        self.tile_cache = plans.load_tile_cache()   # conv geometry -> autotuned (tile id, split-K); optionally persisted
        self._pe = {}
        self._rep = {}
        sd = {k: v.detach().float().cpu() for k, v in sd.items() if v.is_floating_point()}
        self.sd = sd
        self._pe_w = {}
        W, dev = self.w, device
        CS, CV = m['sensory_dim'], m['value_dim']
        up = m['mask_decoder']['up_dims']

        def conv(name, segs=None, bn=None):
            w = sd[name + '.weight']
            b = sd.get(name + '.bias')
            if bn is not None:
                w, b = fold_bn(w, sd, bn)
            W[name] = pack_conv(w, b, dev, segs)

        # ResNet trunks with eval-mode BN folded (resnet.py)
        for prefix, in_ch in (('pixel_encoder', 3), ('mask_encoder', 5)):
            conv(prefix + '.conv1', segs=[(in_ch, 8)], bn=prefix + '.bn1')
            for k in sd:
                if k.startswith(prefix + '.') and k.endswith('.weight') and sd[k].dim() == 4 and k != prefix + '.conv1.weight':
                    name = k[:-len('.weight')]
                    if '.fuser.' in name or '.sensory_update.' in name:
                        continue
                    if name.endswith('.downsample.0'):
                        conv(name, bn=name[:-1] + '1')
                    else:
                        conv(name, bn=name.replace('.conv', '.bn'))
        for name in ['pix_feat_proj', 'key_proj.pix_feat_proj', 'key_proj.key_proj', 'key_proj.d_proj', 'key_proj.e_proj',
                     'mask_decoder.sensory_update.g16_conv', 'mask_decoder.sensory_update.g8_conv',
                     'mask_decoder.decoder_feat_proc.transforms.0', 'mask_decoder.decoder_feat_proc.transforms.1',
                     'mask_decoder.up_16_8.out_conv.downsample', 'mask_decoder.up_16_8.out_conv.conv1',
                     'mask_decoder.up_16_8.out_conv.conv2', 'mask_decoder.up_8_4.out_conv.conv1',
                     'mask_decoder.up_8_4.out_conv.conv2', 'mask_decoder.pred']:
            conv(name)
        conv('mask_decoder.sensory_update.g4_conv', segs=[(up[2], up[2]), (1, 8)])
        # g16_conv(g16) + g8_conv(area(g8)) + g4_conv(area([g4 | logits])) of SensoryUpdater (modules.py:47-61) is ONE 1x1 conv over the
        # virtual concat [g16 | g8' | g4' | logits' (1 -> 64 channels, so every source is a multiple of the 64-channel K tile)]
        sg = 'mask_decoder.sensory_update.'
        W[sg + 'g_all'] = pack_conv(torch.cat([sd[sg + 'g16_conv.weight'], sd[sg + 'g8_conv.weight'], sd[sg + 'g4_conv.weight']], 1),
                                    sd[sg + 'g16_conv.bias'] + sd[sg + 'g8_conv.bias'] + sd[sg + 'g4_conv.bias'], dev,
                                    segs=[(up[0], up[0]), (up[1], up[1]), (up[2], up[2]), (1, 64)])
        conv('mask_decoder.sensory_update.transform', segs=[(CS, CS), (CS, CS)])
        conv('mask_encoder.sensory_update.transform', segs=[(CV, CV), (CS, CS)])
        conv('pixel_fuser.sensory_compress', segs=[(CS, CS), (2, 64)])       # (mask, others) padded to a whole 64-channel K tile: LDS-DMA conv
        ca_blocks = []
        for fz in ('mask_encoder.fuser', 'pixel_fuser.fuser'):
            conv(fz + '.distributor.x_transform')
            conv(fz + '.distributor.g_transform')
            ca_blocks += [fz + '.block1', fz + '.block2']
        t = 'object_transformer'
        ot = m['object_transformer']
        C = m['embed_dim']
        for b in range(ot['num_blocks']):
            q = f'{t}.blocks.{b}'
            ca_blocks.append(q + '.pixel_ffn.conv')
            rp, sa, rq = q + '.read_from_pixel.cross_attn', q + '.self_attn.self_attn', q + '.read_from_query.cross_attn'
            Wp, bp = sd[rp + '.in_proj_weight'], sd[rp + '.in_proj_bias']
            Ws, bs = sd[sa + '.in_proj_weight'], sd[sa + '.in_proj_bias']
            Wq, bq = sd[rq + '.in_proj_weight'], sd[rq + '.in_proj_bias']
            # pixel-side merged projection [k (read_from_pixel) | v (read_from_pixel) | q (read_from_query)]
            wm = torch.cat([Wp[C:2 * C], Wp[2 * C:], Wq[:C]], 0)
            bm = torch.cat([bp[C:2 * C], bp[2 * C:], bq[:C]], 0)
            W[q + '.pixel_proj'] = linear_as_conv(wm, bm, dev)
            # the same projections applied to the (block-invariant) pixel positional term; v gets none
            wpe = torch.cat([Wp[C:2 * C], torch.zeros(C, C), Wq[:C]], 0)
            self._pe_w[b] = wpe
            self._pe_wb = getattr(self, '_pe_wb', {})
            self._pe_wb[b] = wpe.float()
            # ... or inside the block's own projection (round 4): kvq_b = Wm_b pixel_b + Wpe_b (We x + be + PE) as ONE conv over the virtual
            # concat [pixel_b | x] with the composed weights [Wm_b | Wpe_b We]; only the PE term is left as a per-pixel residual (pe_rb).
            # Same flops as one conv x -> [R_0 | R_1 | R_2] in front, without writing and re-reading 3 x 768 channels per pixel and object.
            we_ = sd[t + '.pixel_emb_proj.weight'].float().reshape(C, -1)
            W[q + '.pixel_proj_x'] = pack_conv(torch.cat([wm.float(), wpe.float() @ we_], 1).reshape(3 * C, 2 * C, 1, 1),
                                               bm.float() + wpe.float() @ sd[t + '.pixel_emb_proj.bias'].float(), dev, segs=[(C, C), (C, C)])
            W[q + '.read_from_pixel.q'] = pack_linear(Wp[:C], bp[:C], dev)
            W[q + '.read_from_pixel.out'] = pack_linear(sd[rp + '.out_proj.weight'], sd[rp + '.out_proj.bias'], dev)
            W[q + '.self_attn.qkv'] = pack_linear(Ws, bs, dev)                    # one launch; the query PE feeds q and k only
            W[q + '.self_attn.out'] = pack_linear(sd[sa + '.out_proj.weight'], sd[sa + '.out_proj.bias'], dev)
            W[q + '.read_from_query.kv'] = pack_linear(Wq[C:], bq[C:], dev)        # [k | v]; the query PE feeds k only
            W[q + '.read_from_query.out'] = linear_as_conv(sd[rq + '.out_proj.weight'], sd[rq + '.out_proj.bias'], dev)
            if C == 256:
                W[q + '.read_from_query.out_blob'] = out_proj_blob(W[q + '.read_from_query.out'])       # ATTN_P2Q applies it itself (plans.P2Q_OUT)
            for ln in ('.read_from_pixel.norm', '.self_attn.norm', '.ffn.norm'):
                W[q + ln + '.weight'] = sd[q + ln + '.weight'].to(dev).contiguous()
                W[q + ln + '.bias'] = sd[q + ln + '.bias'].to(dev).contiguous()
            W[q + '.ffn.linear1'] = pack_linear(sd[q + '.ffn.linear1.weight'], sd[q + '.ffn.linear1.bias'], dev)
            W[q + '.ffn.linear2'] = pack_linear(sd[q + '.ffn.linear2.weight'], sd[q + '.ffn.linear2.bias'], dev)
        # the positional term R_b = [Wk_b pe | 0 | Wq2_b pe] of every block depends only on pixel_pe: one conv for all blocks
        wpe_all = torch.cat([self._pe_w.pop(b) for b in range(ot['num_blocks'])], 0).float()        # [3C*nb, C]
        self._wpe_all = wpe_all
        W[t + '.pe_proj_all'] = linear_as_conv(wpe_all, None, dev)
        # ... and since pixel_pe = pixel_emb_proj(x) + PE is consumed by nothing else, the two 1x1 maps are composed (in fp32, once):
        # R_all = (Wpe We) x + Wpe be + Wpe PE.  One conv x -> [pixel | R_all]; the PE term is a per-pixel residual (Engine.pe_r)
        we = sd[t + '.pixel_emb_proj.weight'].float().reshape(C, -1)
        W[t + '.pixel_init_R'] = pack_conv(torch.cat([sd[t + '.pixel_init_proj.weight'].float(), (wpe_all @ we).reshape(-1, we.shape[1], 1, 1)], 0),
                                           torch.cat([sd[t + '.pixel_init_proj.bias'].float(), wpe_all @ sd[t + '.pixel_emb_proj.bias'].float()], 0), dev)
        conv(t + '.pixel_init_proj')
        # pixel_init_proj | pixel_emb_proj read the same input: one conv with 2C output channels
        W[t + '.pixel_init_emb'] = pack_conv(torch.cat([sd[t + '.pixel_init_proj.weight'], sd[t + '.pixel_emb_proj.weight']], 0),
                                             torch.cat([sd[t + '.pixel_init_proj.bias'], sd[t + '.pixel_emb_proj.bias']], 0), dev)
        for b in range(ot['num_blocks'] + 1):
            conv(f'{t}.mask_pred.{b}.1')
        for name in ca_blocks:
            conv(name + '.conv1')
            conv(name + '.conv2')
            W[name + '.conv.weight'] = sd[name + '.conv.weight'].reshape(-1).to(dev).contiguous()
        W[t + '.summary_to_query_init'] = pack_linear(sd[t + '.summary_to_query_init.weight'], sd[t + '.summary_to_query_init.bias'], dev)
        W[t + '.summary_to_query_emb'] = pack_linear(sd[t + '.summary_to_query_emb.weight'], sd[t + '.summary_to_query_emb.bias'], dev)
        u = 'object_summarizer'
        for name in ('.input_proj', '.feature_pred.0', '.feature_pred.2', '.weights_pred.0', '.weights_pred.2'):
            W[u + name] = linear_as_conv(sd[u + name + '.weight'], sd[u + name + '.bias'], dev)
        # The summarizer's five per-pixel linears as TWO launches (round 5; object_summarizer.py:55-89).  input_proj feeds nothing but the
        # first layers of the two MLPs (feature_pred.0, weights_pred.0), and the positional encoding is added in between, so the maps are
        # composed in fp32: [f1 | w1] = relu(W0 W_in value + W0 (b_in + PE) + b0), W0 = [Wf0 ; Ww0] -- one conv value -> 2C channels with the
        # PE term as a per-pixel broadcast residual (Engine.pe_sum).  The second layers are one block-diagonal conv [f1 | w1] ->
        # [feature (C) | weight logits (Q)], stored in fp32 (the reference's fp32 island), read by SUMMARIZE with a row stride.
        w_in, b_in = sd[u + '.input_proj.weight'].float(), sd[u + '.input_proj.bias'].float()
        w0 = torch.cat([sd[u + '.feature_pred.0.weight'].float(), sd[u + '.weights_pred.0.weight'].float()], 0)
        b0 = torch.cat([sd[u + '.feature_pred.0.bias'].float(), sd[u + '.weights_pred.0.bias'].float()], 0)
        self._sum_w0 = w0
        W[u + '.in_fw0'] = linear_as_conv(w0 @ w_in, b0 + w0 @ b_in, dev)
        wf2, ww2 = sd[u + '.feature_pred.2.weight'].float(), sd[u + '.weights_pred.2.weight'].float()
        cf, cw = wf2.shape[1], ww2.shape[1]
        w2 = torch.zeros(wf2.shape[0] + ww2.shape[0], cf + cw)
        w2[:wf2.shape[0], :cf] = wf2
        w2[wf2.shape[0]:, cf:] = ww2
        W[u + '.fw2'] = linear_as_conv(w2, torch.cat([sd[u + '.feature_pred.2.bias'].float(), sd[u + '.weights_pred.2.bias'].float()], 0), dev)

    def pe(self, h, w):
        """bf16 [h*w, C] positional encoding (both PositionalEncoding instances use the same formula)."""
        if (h, w) not in self._pe:
            e = plans.positional_encoding(h, w, self.m['embed_dim'], self.m['pixel_pe_scale'], self.m['pixel_pe_temperature'])
            self._pe[(h, w)] = e.reshape(h * w, -1).to(BF16).to(self.device).contiguous()
        return self._pe[(h, w)]

    def pe0(self, h, w):
        """bf16 [h*w, 2C]: [0 | positional encoding] -- the broadcast residual of the merged pixel_init | pixel_emb conv."""
        key = ('pe0', h, w)
        if key not in self._pe:
            e = self.pe(h, w)
            self._pe[key] = torch.cat([torch.zeros_like(e), e], 1).contiguous()
        return self._pe[key]

    def pe_r(self, h, w):
        """bf16 [h*w, C + 3C*blocks]: [0 | Wpe PE] -- the broadcast residual of the composed pixel_init | R_all conv."""
        key = ('pe_r', h, w)
        if key not in self._pe:
            e = plans.positional_encoding(h, w, self.m['embed_dim'], self.m['pixel_pe_scale'], self.m['pixel_pe_temperature'])
            e = e.reshape(h * w, -1).float()
            r = e @ self._wpe_all.t()
            self._pe[key] = torch.cat([torch.zeros_like(e), r], 1).to(BF16).to(self.device).contiguous()
        return self._pe[key]

    def pe_rb(self, h, w, b):
        """bf16 [h*w, 3C]: Wpe_b PE -- the broadcast residual of block b's composed pixel projection (Engine: '.pixel_proj_x')."""
        key = ('pe_rb', h, w, b)
        if key not in self._pe:
            e = plans.positional_encoding(h, w, self.m['embed_dim'], self.m['pixel_pe_scale'], self.m['pixel_pe_temperature'])
            e = e.reshape(h * w, -1).float()
            self._pe[key] = (e @ self._pe_wb[b].t()).to(BF16).to(self.device).contiguous()
        return self._pe[key]

    def pe_sum(self, h, w):
        """bf16 [h*w, 2C]: [Wf0 ; Ww0] PE -- the broadcast residual of the summarizer's composed first conv (Engine: '.in_fw0')."""
        key = ('pe_sum', h, w)
        if key not in self._pe:
            e = plans.positional_encoding(h, w, self.m['embed_dim'], self.m['pixel_pe_scale'], self.m['pixel_pe_temperature'])
            self._pe[key] = (e.reshape(h * w, -1).float() @ self._sum_w0.t()).to(BF16).to(self.device).contiguous()
        return self._pe[key]

    def rep_embedding(self, which, K):
        key = (which, K)
        if key not in self._rep:
            e = self.sd[f'object_transformer.{which}.weight']
            self._rep[key] = e.repeat(K, 1).to(self.device).contiguous()
        return self._rep[key]

    def mask_down_bufs(self, K, h, w, clips=1):
        """(pair bf16 [K, h, w, 64] -- channels 0, 1 = (mask, others), the rest stays zero --, m16 f32 [K, h, w]): MASK_DOWN results that
        the decoder's last launch of one frame leaves for the pixel fusion of the next (plans.build_segment(md) / build_pixel_fusion(pre_md)).
        clips > 1: the buffers of the lock-step plans (K = clips x objects per clip), a set of their own."""
        mb = self.__dict__.setdefault('_md_bufs', {})
        if (K, h, w, clips) not in mb:
            mb[(K, h, w, clips)] = (torch.zeros((K, h, w, 64), dtype=BF16, device=self.device), torch.zeros((K, h, w), dtype=F32, device=self.device))
        return mb[(K, h, w, clips)]

    def query_bufs(self, K):
        """f32 [K * Q, C] x 2: the initial object queries / query embeddings of the transformer, shared by the plan variants that write
        and that only read them (CUTIE.readout_query); outside the activation arena, so they survive between frames."""
        qb = self.__dict__.setdefault('_query_bufs', {})
        if K not in qb:
            M = K * self.m['object_transformer']['num_queries']
            qb[K] = tuple(torch.zeros((M, self.m['embed_dim']), dtype=F32, device=self.device) for _ in range(2))
        return qb[K]

    def fork(self):
        """A second engine over the SAME packed weights and autotune cache with its own plans (= its own activation
        buffers): what a second clip processed concurrently on another stream needs."""
        e = object.__new__(Engine)
        e.__dict__.update(self.__dict__)
        e._plans = {}
        e._arenas = {}
        e.pool = plans.SlotPool()
        e.__dict__.pop('_splitk_part', None)
        e.__dict__.pop('_streams', None)            # (the look-ahead streams of InferenceCore: one set per engine)
        e.__dict__.pop('_query_bufs', None)
        e.__dict__.pop('_md_bufs', None)
        e.__dict__.pop('_qinit_state', None)
        return e

    def plan(self, key, builder, *args):
        p = self._plans.get(key)
        if p is None:
            p = builder(self, *args)
            p.ol.finalize()
            if plans.ARENA:
                # the image encoder (+ key projection) may run on the look-ahead stream next to everything else: its own arena
                kind = 'side' if key[0] in ('enc', 'key') else ('win' if key[0] == 'encw' else 'main')     # (the window encoder: a third stream)
                arenas = self.__dict__.setdefault('_arenas', {})
                if kind not in arenas:
                    arenas[kind] = plans.Arena(self.device)
                p.pack_into(arenas[kind])
            self._plans[key] = p
        return p


class _Container(nn.Module):
    pass


class CUTIE(nn.Module):
    def __init__(self, cfg, *, single_object=False):
        super().__init__()
        if single_object:
            raise NotImplementedError('single_object=True is a training-time variant; not on the inference path')
        self.cfg = cfg
        m = cfg.model if hasattr(cfg, 'model') else cfg['model']
        self.model_cfg = m
        self.ms_dims = m['pixel_encoder']['ms_dims']
        self.key_dim, self.value_dim = m['key_dim'], m['value_dim']
        self.sensory_dim, self.pixel_dim, self.embed_dim = m['sensory_dim'], m['pixel_dim'], m['embed_dim']
        self.single_object = single_object
        self.object_transformer_enabled = m['object_transformer']['num_blocks'] > 0
        if not self.object_transformer_enabled:
            raise NotImplementedError('num_blocks == 0 is not supported')
        self._spec = build_spec(m)
        g = torch.Generator().manual_seed(0)
        for name, (shape, role) in self._spec.items():
            self._register(name, shape, role, g)
        self.register_buffer('pixel_mean', torch.tensor(m['pixel_mean'], dtype=F32).view(-1, 1, 1), False)
        self.register_buffer('pixel_std', torch.tensor(m['pixel_std'], dtype=F32).view(-1, 1, 1), False)
        self._eng = None
        self._key_cache = None
        self.eval()

    # ---- parameter tree with the reference's names --------------------------------------------------
    def _register(self, name, shape, role, g):
        parts = name.split('.')
        mod = self
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Container())
            mod = getattr(mod, p)
        leaf = parts[-1]
        if role in ('w', 'emb', 'eca'):
            fan_in = max(1, int(torch.tensor(shape[1:]).prod())) if len(shape) > 1 else 1
            t = torch.randn(shape, generator=g) * (1.0 / fan_in ** 0.5)
            mod.register_parameter(leaf, nn.Parameter(t))
        elif role in ('b', 'bn_b', 'ln_b'):
            mod.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))
        elif role in ('bn_w', 'ln_w'):
            mod.register_parameter(leaf, nn.Parameter(torch.ones(shape)))
        elif role == 'bn_mean':
            mod.register_buffer(leaf, torch.zeros(shape))
        elif role == 'bn_var':
            mod.register_buffer(leaf, torch.ones(shape))
        elif role == 'bn_nbt':
            mod.register_buffer(leaf, torch.tensor(0, dtype=torch.long))
        elif role == 'buf':
            dim = shape[0] * 2
            inv = 1.0 / (self.model_cfg['pixel_pe_temperature'] ** (torch.arange(0, dim, 2).float() / dim))
            mod.register_buffer(leaf, inv)
        else:
            raise KeyError(role)

    def train(self, mode=True):
        # inference-only module; BN statistics are always frozen in the reference too (big_modules.py:56-61)
        return super().train(False)

    def _apply(self, fn, *a, **k):
        self._eng = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._eng = None
        return super().load_state_dict(*a, **k)

    def load_weights(self, src_dict, init_as_zero_if_needed=False) -> None:
        """cutie.py:212-256: accepts single-object checkpoints (4-channel mask_encoder.conv1 etc.)."""
        src_dict = dict(src_dict)
        k = 'mask_encoder.conv1.weight'
        if k in src_dict and src_dict[k].shape[1] == 4:
            log.info(f'Converting {k} from single object to multiple objects.')
            pads = torch.zeros((64, 1, 7, 7), device=src_dict[k].device)
            if not init_as_zero_if_needed:
                nn.init.orthogonal_(pads)
            src_dict[k] = torch.cat([src_dict[k], pads], 1)
        k = 'pixel_fuser.sensory_compress.weight'
        if k in src_dict and src_dict[k].shape[1] == self.sensory_dim + 1:
            log.info(f'Converting {k} from single object to multiple objects.')
            pads = torch.zeros((self.value_dim, 1, 1, 1), device=src_dict[k].device)
            if not init_as_zero_if_needed:
                nn.init.orthogonal_(pads)
            src_dict[k] = torch.cat([src_dict[k], pads], 1)
        own = self.state_dict()
        for k in src_dict:
            if k not in own:
                log.info(f'Key {k} found in src_dict but not in self.state_dict()!!!')
        for k in own:
            if k not in src_dict:
                log.info(f'Key {k} found in self.state_dict() but not in src_dict!!!')
        self.load_state_dict(src_dict, strict=False)

    @property
    def device(self) -> torch.device:
        d = self.__dict__.get('_device_cache')     # (asked a dozen times per frame; nn.Module's buffer look-up is ~1 us)
        if d is None:
            d = self.__dict__['_device_cache'] = self.pixel_mean.device
        return d

    def _apply(self, fn, *args, **kwargs):         # .cuda() / .to() / .cpu(): the module may have moved
        self.__dict__.pop('_device_cache', None)
        return super()._apply(fn, *args, **kwargs)

    def fork(self) -> 'CUTIE':
        """A view of this network for a second concurrent clip (own HIP stream / host thread): shares the parameters, the
        packed device weights and the autotuned tile choices, owns its launch plans and activation buffers.
        (No counterpart in the reference, where torch allocates activations per call.)"""
        import copy
        f = copy.copy(self)                      # nn.Module shallow copy: the parameter / buffer dicts are shared
        f._eng = self.engine().fork()
        f._key_cache = None
        return f

    def engine(self) -> Engine:
        if self._eng is None:
            ex = _lib.get_executor()      # fails loudly when the HIP library / device is missing
            if not ex.is_mock and self.device.type != 'cuda':
                raise RuntimeError('CUTIE parameters live on %s: move the module to the MI355X first (.cuda()); '
                                   'cutie_amd has no CPU path' % self.device)
            self._eng = Engine(self.state_dict(), self.model_cfg, self.device)
        return self._eng

    # ---- facade methods ---------------------------------------------------------------------------------
    def _encode(self, image, h0, w0, H, W, pad_left, pad_top):
        """image f32 [3,h0,w0] (un-padded) -> dict of physical outputs (fused encode_image + transform_key)."""
        eng = self.engine()
        dev = eng.device
        m = self.model_cfg
        P = eng.plan(('enc', h0, w0, H, W, pad_left, pad_top), plans.build_encode, h0, w0, H, W, pad_left, pad_top)
        h, w = H // 16, W // 16
        # Outputs from the frame-slot pool (stable addresses: the plan replays as a HIP graph).  The query-side affinity operands
        # Bhi / Blo / cq have padding rows [hw, HWp) that must stay zero and are never written: every pooled tensor starts zeroed.
        cached = eng.enc_specs.get((h, w))
        if cached is None:
            hw = h * w
            HWp = -(-hw // 64) * 64
            ms = self.ms_dims
            W_ = eng.w
            Z = False
            specs = dict(f16=((1, h, w, ms[0]), BF16, Z), f8=((1, 2 * h, 2 * w, ms[1]), BF16, Z), f4=((1, 4 * h, 4 * w, ms[2]), BF16, Z),
                         pix_feat=((1, h, w, m['pixel_dim']), BF16, Z), key=((hw, m['key_dim']), F32, Z), shr=((hw,), F32, Z),
                         sel=((hw, m['key_dim']), F32, Z), Bhi=((HWp, 128), BF16, True), Blo=((HWp, 128), BF16, True), cq=((HWp,), F32, True),
                         f8p=((1, 2 * h, 2 * w, W_['mask_decoder.decoder_feat_proc.transforms.0'].cout), BF16, Z),
                         f4p=((1, 4 * h, 4 * w, W_['mask_decoder.decoder_feat_proc.transforms.1'].cout), BF16, Z),
                         fuse_xt=((1, h, w, W_['pixel_fuser.fuser.distributor.x_transform'].cout), BF16, Z))
            cached = eng.enc_specs[(h, w)] = (('enc', h, w, eng.devstr), specs)
        pool_key, specs = cached
        out = eng.pool.get(pool_key, specs, dev)
        P.graph_head = 1                                       # IMG_PREP reads the caller's frame
        image = f32c(image)
        P.run(image=image, **out)
        out['h'], out['w'] = h, w
        # image-only results that the decoder / pixel fuser consume (found by the address of the feature they derive from)
        frame_context.remember('decoder_feats', out['f8'], (out['f8p'], out['f4p'], out['f4']), cap=4)
        frame_context.remember('fuse_xt', out['pix_feat'], out['fuse_xt'], cap=4)
        return out

    def _encode_window(self, images, h0, w0, H, W, pad_left, pad_top, qrows=64):
        """B un-padded frames f32 [3,h0,w0] of one geometry through ONE plan (plans.build_encode(B=...)) -> a list of B records shaped
        like `_encode`'s (views of the [B, ...] outputs).  Frame b's record is bit-identical to `_encode(images[b], ...)`.  The
        frame_context entries of a record are NOT made here: `_adopt_encoded` registers them when the frame is about to be consumed
        (the tables keep the last few frames only).  qrows: see plans.build_encode."""
        eng = self.engine()
        dev = eng.device
        m = self.model_cfg
        B = len(images)
        P = eng.plan(('encw', B, h0, w0, H, W, pad_left, pad_top, qrows), plans.build_encode, h0, w0, H, W, pad_left, pad_top, B, qrows)
        h, w = H // 16, W // 16
        hw = h * w
        HWp = -(-hw // qrows) * qrows
        ms = self.ms_dims
        W_ = eng.w
        Z = False
        specs = dict(f16=((B, h, w, ms[0]), BF16, Z), f8=((B, 2 * h, 2 * w, ms[1]), BF16, Z), f4=((B, 4 * h, 4 * w, ms[2]), BF16, Z),
                     pix_feat=((B, h, w, m['pixel_dim']), BF16, Z), key=((B, hw, m['key_dim']), F32, Z), shr=((B, hw), F32, Z),
                     sel=((B, hw, m['key_dim']), F32, Z), Bhi=((B, HWp, 128), BF16, True), Blo=((B, HWp, 128), BF16, True), cq=((B, HWp), F32, True),
                     f8p=((B, 2 * h, 2 * w, W_['mask_decoder.decoder_feat_proc.transforms.0'].cout), BF16, Z),
                     f4p=((B, 4 * h, 4 * w, W_['mask_decoder.decoder_feat_proc.transforms.1'].cout), BF16, Z),
                     fuse_xt=((B, h, w, W_['pixel_fuser.fuser.distributor.x_transform'].cout), BF16, Z))
        out = eng.pool.get_ring(('encw', B, h, w, HWp, eng.devstr), specs, dev, ring=4)
        P.run(**({'image': images[0]} if B == 1 else {'image%d' % b: images[b] for b in range(B)}), **out)
        recs = []
        for b in range(B):
            o = {k: (v[b:b + 1] if k in ('f16', 'f8', 'f4', 'pix_feat', 'f8p', 'f4p', 'fuse_xt') else v[b]) for k, v in out.items()}
            o['h'], o['w'] = h, w
            recs.append(o)
        return recs

    def _adopt_encoded(self, o):
        """A record of `_encode_window` -> what `_encode_image_raw` + `transform_key` return for that frame, with the companions of
        the results registered (decoder / fuser convolutions, similarity operands of the key)."""
        frame_context.remember('decoder_feats', o['f8'], (o['f8p'], o['f4p'], o['f4']), cap=4)
        frame_context.remember('fuse_xt', o['pix_feat'], o['fuse_xt'], cap=4)
        ms = (logical(o['f16']), logical(o['f8']), logical(o['f4']))
        key, shr, sel = self._key_views(o)
        qo = o.get('_qo')
        if qo is None:
            qo = o['_qo'] = dict(Bhi=o['Bhi'], Blo=o['Blo'], cq=o['cq'], h=o['h'], w=o['w'])
        frame_context.remember('query', o['key'], qo, cap=4)
        return ms, logical(o['pix_feat']), key, shr, sel

    def encode_image(self, image: torch.Tensor) -> (Iterable[torch.Tensor], torch.Tensor):
        """image [1,3,H,W] in [0,1] (already padded to /16) -> ((f16, f8, f4), pix_feat); cutie.py:61-64"""
        assert image.dim() == 4 and image.shape[0] == 1, 'batch size 1 only'
        H, W = image.shape[-2:]
        return self._encode_image_raw(image[0], H, W, H, W, 0, 0)

    def _encode_image_raw(self, image, h0, w0, H, W, pad_left, pad_top):
        """Same as encode_image for an un-padded frame [3,h0,w0]: the zero pad of
        InferenceCore.step (tensor_utils.pad_divide_by) is fused into the first kernel."""
        o = self._encode(image, h0, w0, H, W, pad_left, pad_top)
        ms = (logical(o['f16']), logical(o['f8']), logical(o['f4']))
        self._key_cache = (o['f16'], o)
        return ms, logical(o['pix_feat'])

    def _key_views(self, o):
        h, w = o['h'], o['w']
        key = o['key'].view(1, h, w, -1).permute(0, 3, 1, 2)
        shr = o['shr'].view(1, 1, h, w)
        sel = o['sel'].view(1, h, w, -1).permute(0, 3, 1, 2)
        return key, shr, sel

    def transform_key(self, final_pix_feat: torch.Tensor, *, need_sk: bool = True, need_ek: bool = True):
        """cutie.py:92-98.  Returns fp32 (key, shrinkage, selection) in logical [1,C,h,w] shapes."""
        f16 = nhwc_of(final_pix_feat)
        if self._key_cache is not None and self._key_cache[0].data_ptr() == f16.data_ptr():
            o = self._key_cache[1]            # computed by the fused encode plan
        else:
            eng = self.engine()
            h, w = f16.shape[1:3]
            hw, HWp = h * w, -(-h * w // 64) * 64
            dev = self.device
            o = dict(key=torch.empty((hw, self.key_dim), dtype=F32, device=dev), shr=torch.empty((hw,), dtype=F32, device=dev),
                     sel=torch.empty((hw, self.key_dim), dtype=F32, device=dev),
                     Bhi=torch.zeros((HWp, 128), dtype=BF16, device=dev), Blo=torch.zeros((HWp, 128), dtype=BF16, device=dev),
                     cq=torch.zeros((HWp,), dtype=F32, device=dev), h=h, w=w)
            P = eng.plan(('key', h, w), plans.build_transform_key, h, w)
            P.run(f16=f16, **{k: v for k, v in o.items() if k not in ('h', 'w')})
        key, shr, sel = self._key_views(o)
        # the similarity operands of this key, found again by MemoryManager.read: ONLY what the read-out needs (the whole encoder output
        # would stay pinned -- ~40 MB per 480p frame -- for as long as the entry lives), and only the last few frames
        qo = o.get('_qo')
        if qo is None:
            qo = o['_qo'] = dict(Bhi=o['Bhi'], Blo=o['Blo'], cq=o['cq'], h=o['h'], w=o['w'])
        frame_context.remember('query', o['key'], qo, cap=4)
        return key, (shr if need_sk else None), (sel if need_ek else None)

    def query_operands(self, query_key: torch.Tensor, selection: torch.Tensor) -> dict:
        """Split-bf16 similarity operands (Bhi, Blo, cq) of a query key.  Fast path: the key came out of ``transform_key`` (or is a
        view of it) and its operands were computed by the same launch plan.  Slow path (a caller that cloned / rebuilt the key, or
        brings its own): one KEY_PREP launch on fp32 copies of key and selection."""
        o = frame_context.recall('query', query_key)
        if o is not None:
            return o
        assert query_key.dim() == 4 and query_key.shape[0] == 1 and selection is not None, 'query key [1,CK,h,w] with its selection'
        h, w = query_key.shape[-2:]
        hw, HWp = h * w, -(-h * w // 64) * 64
        dev = self.device
        kphys = query_key[0].permute(1, 2, 0).reshape(hw, -1).to(device=dev, dtype=F32).contiguous()
        ephys = selection[0].permute(1, 2, 0).reshape(hw, -1).to(device=dev, dtype=F32).contiguous()
        o = dict(key=kphys, sel=ephys, Bhi=torch.zeros((HWp, 128), dtype=BF16, device=dev), Blo=torch.zeros((HWp, 128), dtype=BF16, device=dev),
                 cq=torch.zeros((HWp,), dtype=F32, device=dev), h=h, w=w)
        from .. import ops as O
        ol = O.OpList()
        ol.key_prep(kphys, ephys, o['Bhi'], o['Blo'], o['cq'], n=hw, query=True)
        ol.run()
        frame_context.remember('query', query_key, o, cap=4)
        return o

    @staticmethod
    def _sensory_pair(sensory):
        """sensory logical [1,K,CS,h,w] fp32 -> (phys f32 [K,h,w,CS], bf16 shadow)."""
        phys = group_nhwc_of(sensory, F32)
        shadow = frame_context.recall('sensory_bf16', phys)
        if shadow is None or shadow.shape != phys.shape:
            shadow = phys.to(BF16)                # a state the HIP path has not produced itself: one cast
        return phys, shadow

    def encode_mask(self, image, ms_features, sensory, masks, *, deep_update=True, chunk_size=-1, need_weights=False,
                    _raw=None, _split=False):
        """cutie.py:66-90.  image [1,3,H,W]; ms_features = stride-16 pix_feat; sensory [1,K,CS,h,w] fp32 (updated in
        place when deep_update); masks [1,K,H,W] -> (value, sensory, summaries [1,K,Q,C+1], None)."""
        eng = self.engine()
        dev = eng.device
        K = masks.shape[1]
        H, W = masks.shape[-2:]
        h, w = H // 16, W // 16
        if _raw is not None:
            img, h0, w0, pl, pt = _raw
        else:
            img, h0, w0, pl, pt = image[0], H, W, 0, 0
        pix = nhwc_of(ms_features)
        sf, sb = self._sensory_pair(sensory)
        mk = f32c(masks[0])
        # MASK_DOWN(masks) is already there when these masks are the probabilities the last segment() returned (every memory frame of a
        # propagation): its up-sampling launch left them for the next frame's pixel fusion (see segment / pixel_fusion)
        md = frame_context.recall('mask_down', mk)
        md = md is not None and md == (K, h, w, eng.__dict__.get('_md_gen')) and plans.SUM_FUSED and not plans.UNFUSED
        P = eng.plan(('emask', K, h0, w0, H, W, pl, pt, bool(deep_update), md), plans.build_encode_mask, K, h0, w0, H, W, pl, pt,
                     bool(deep_update), md)
        o = eng.pool.get(('emask', K, h, w, eng.devstr),
                         dict(value=((K, h, w, self.value_dim), BF16, False),
                              summ=((K, self.model_cfg['object_summarizer']['num_summaries'], self.embed_dim + 1), F32, False)), dev)
        value, summ = o['value'], o['summ']
        dyn = dict(image=f32c(img), masks=mk, pix_feat=pix, sensory_f32=sf, sensory_bf16=sb, value=value, summ=summ)
        if _split:
            # two parts: the mask values first (what the memory bank needs), the sensory deep update + object summaries when the caller
            # asks for them -- InferenceCore inserts the values and starts the NEXT frame's affinity read-out on its side stream in between
            cut = P.meta['value_done']
            P.run_part(0, cut, **dyn)

            def finish():
                P.run_part(cut, None, first=False, **dyn)
                frame_context.remember('sensory_bf16', sf, sb)
                return group_logical(sf), summ.unsqueeze(0)
            return group_logical(value), finish
        P.run(**dyn)
        new_sens = group_logical(sf)
        frame_context.remember('sensory_bf16', sf, sb)
        return group_logical(value), new_sens, summ.unsqueeze(0), None

    def pixel_fusion(self, pix_feat, pixel, sensory, last_mask, *, chunk_size=-1):
        """cutie.py:142-157.  pixel = memory readout [1,K,CV,h,w]; last_mask [1,K,H,W] -> fused [1,K,CE,h,w]"""
        eng = self.engine()
        pf = nhwc_of(pix_feat)
        px = group_nhwc_of(pixel)
        K, h, w = px.shape[:3]
        _, sb = self._sensory_pair(sensory)
        lm = f32c(last_mask[0])
        xt = None if plans.UNFUSED else frame_context.recall('fuse_xt', pf)             # x_transform(pix_feat), computed with the encoder (None: a caller's own features)
        md = frame_context.recall('mask_down', lm)          # MASK_DOWN(last_mask), left by the segment() that produced this very tensor ...
        md = md is not None and md == (K, h, w, eng.__dict__.get('_md_gen'))      # ... if no later segment() has overwritten it
        P = eng.plan(('fuse', K, h, w, xt is not None, md), plans.build_pixel_fusion, K, h, w, xt is not None, md)
        fused = eng.pool.get(('fuse', K, h, w, eng.devstr), dict(fused=((K, h, w, self.embed_dim), BF16, False)), eng.device)['fused']
        P.run(pix_feat=pf, pixel=px, sensory_bf16=sb, last_mask=lm, fused=fused, **({} if xt is None else {'fuse_xt': xt}))
        return group_logical(fused)

    def readout_query(self, pixel_readout, obj_memory, *, selector=None, need_weights=False, _last_aux=True, _summary_token=None):
        """cutie.py:159-170 -> QueryTransformer.  obj_memory [1,K,T,Q,C+1] (T summed) -> (pixel [1,K,C,h,w], aux).
        _last_aux=False (MemoryManager, unless save_aux): the logits of the last block are not computed (aux['logits'] has one
        entry less); they do not influence the read-out."""
        assert selector is None, 'selector is a training-time argument'
        eng = self.engine()
        px = group_nhwc_of(pixel_readout)
        K, h, w = px.shape[:3]
        om = obj_memory[0]
        if om.dtype != F32:
            om = om.to(F32)
        om = om.sum(dim=1) if om.shape[1] != 1 else om[:, 0]
        if not om.is_contiguous():
            om = om.contiguous()
        # _summary_token (MemoryManager): identifies the CONTENT of obj_memory.  The query initialisation (summaries -> queries, two
        # linears) depends on nothing else; while the token repeats, the plan variant without that launch runs on the queries of the
        # last call (object_transformer.py:118-127 recomputes them every frame; they change on memory frames only).
        st = eng.__dict__.setdefault('_qinit_state', {})
        fresh = _summary_token is None or st.get(K) != _summary_token or not plans.QINIT_SKIP
        st[K] = _summary_token
        P = eng.plan(('rq', K, h, w, bool(_last_aux), fresh), plans.build_readout_query, K, h, w, bool(_last_aux), fresh)
        out = eng.pool.get(('rq', K, h, w, eng.devstr), dict(out=((K, h, w, self.embed_dim), BF16, False)), eng.device)['out']
        P.run(pixel=px, obj_mem=om, out=out)
        n_aux = P.bufs['aux_logits'].shape[0] - (0 if _last_aux else 1)
        aux = {'logits': [P.bufs['aux_logits'][i].view(1, K, h, w) for i in range(n_aux)],
               'q_weights': None, 'p_weights': None}
        return group_logical(out), aux

    def segment(self, ms_image_feat: List[torch.Tensor], memory_readout, sensory, *, selector=None, chunk_size=-1,
                update_sensory=True, _need_logits=True, _fork=False):
        """cutie.py:172-203.  -> (sensory [1,K,CS,h,w], logits [1,K+1,H,W], prob [1,K+1,H,W]) fp32
        _fork (InferenceCore, when the caller announces its next frames): the softmax launch on an auxiliary stream, see below."""
        assert selector is None, 'selector is a training-time argument'
        eng = self.engine()
        dev = eng.device
        p16 = group_nhwc_of(memory_readout)
        K, h, w = p16.shape[:3]
        f8, f4 = nhwc_of(ms_image_feat[1]), nhwc_of(ms_image_feat[2])
        sf, sb = self._sensory_pair(sensory)
        pre = None if plans.UNFUSED else frame_context.recall('decoder_feats', f8)     # decoder_feat_proc(f8, f4), computed with the encoder
        if pre is not None and pre[2].data_ptr() != f4.data_ptr():
            pre = None                                         # (f8 of one frame with f4 of another: a caller's own mix)
        # (only on the caller's announcement: its contract -- do not modify handed-out tensors between steps -- is what makes the
        # MASK_DOWN computed here valid for the next frame; inference tensors carry no version counter to check it with)
        md = bool(_fork) and plans.SEG_MD and not plans.UNFUSED and K + 1 <= 8 and dev.type == 'cuda'
        P = eng.plan(('seg', K, h, w, bool(update_sensory), pre is not None, md), plans.build_segment, K, h, w, bool(update_sensory), pre is not None, md)
        sp = dict(prob=((K + 1, 16 * h, 16 * w), F32, False))
        if _need_logits:
            sp['lup'] = ((K + 1, 16 * h, 16 * w), F32, False)
        o = eng.pool.get(('seg', K, h, w, bool(_need_logits), eng.devstr), sp, dev)   # (a caller that keeps the probabilities keeps the slot: see SlotPool)
        prob, lup = o['prob'], o.get('lup')
        feats = dict(f8=f8, f4=f4) if pre is None else dict(f8p=pre[0], f4p=pre[1])
        dyn = dict(p16=p16, sensory_f32=sf, sensory_bf16=sb, prob=prob, logits_up=lup, **feats)
        n_ops, cut = len(P.ol.recs), P.meta.get('logits_done', 0)
        if _fork and plans.SEG_FORK and not (plans.ONE_LANE or eng.one_lane) and update_sensory and dev.type == 'cuda' and not plans.GRAPHS and 0 < cut < n_ops - 1 and not plans.UNFUSED and K + 1 <= 16:
            # Behind the logits the plan forks: [area pooling, two convs, GRU] update the sensory state, the LAST launch up-samples the
            # logits and takes the softmax.  Neither branch reads what the other writes, and every launch is a serial step of the
            # frame's critical path (~3.4 us of launch boundary on top of its run time): the softmax launch goes to an auxiliary
            # stream of the engine, the caller's stream takes it back in behind the sensory update.  Same launches, same inputs.
            # Measured in one box (tools/r4_call18.sh): +1.1 % with look-ahead hints, -4 % without them (there the frame is one
            # stream and the host's extra calls cost more than the overlap returns) -- hence only on the caller's request.
            main = torch.cuda.current_stream(dev)
            st = eng.__dict__.setdefault('_streams', {})
            if 'aux' not in st:
                st['aux'] = torch.cuda.Stream(device=dev)
            aux = st['aux']
            P.run_part(0, cut, **dyn)
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                P.run_part(n_ops - 1, n_ops, first=False)
            P.run_part(cut, n_ops - 1, first=False)
            main.wait_stream(aux)
        else:
            P.run(**dyn)
        new_sens = group_logical(sf)
        frame_context.remember('sensory_bf16', sf, sb)
        if md:
            eng._md_gen = eng.__dict__.get('_md_gen', 0) + 1
            frame_context.remember('mask_down', prob[1:], (K, h, w, eng._md_gen), cap=2)   # found again by pixel_fusion through last_mask = prob[1:]
        else:
            frame_context.forget('mask_down', prob[1:])                           # (the slot may carry the entry of an earlier frame)
        return new_sens, (lup.unsqueeze(0) if lup is not None else None), prob.unsqueeze(0)

    def read_memory(self, *a, **k):
        raise NotImplementedError('read_memory is the training-time path (cutie.py:102-140); inference reads through MemoryManager')

    def compute_aux(self, *a, **k):
        raise NotImplementedError('aux heads are training-only (cutie.py:205-207)')

    def forward(self, *args, **kwargs):
        raise NotImplementedError
