"""Oracle network: torch-fp32 CPU restatement of the reference ``CUTIE`` facade.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style over a flat
weight dict; tensor layouts follow the reference (NCHW, objects on dim 1).
Every method cites the reference file:line it follows.
"""
import math
import torch
import torch.nn.functional as F

from .weights import MODEL_CFG


def aggregate(prob, dim):
    # cutie/utils/tensor_utils.py:47-55
    prob = prob.float()
    new_prob = torch.cat([torch.prod(1 - prob, dim=dim, keepdim=True), prob], dim).clamp(1e-7, 1 - 1e-7)
    return torch.log(new_prob / (1 - new_prob))


def positional_encoding(h, w, dim_total, scale, temperature):
    """Sinusoidal PE, returns [h, w, dim_total] (x half first, then y half).
    cutie/model/transformer/positional_encoding.py:20-97 (normalize=True, eps=1e-6)."""
    dim = int(math.ceil(dim_total / 4) * 2)
    inv_freq = 1.0 / (temperature ** (torch.arange(0, dim, 2).float() / dim))
    pos_y = torch.arange(h, dtype=torch.float32)
    pos_x = torch.arange(w, dtype=torch.float32)
    pos_y = pos_y / (pos_y[-1] + 1e-6) * scale
    pos_x = pos_x / (pos_x[-1] + 1e-6) * scale
    sy = pos_y[:, None] * inv_freq[None, :]
    sx = pos_x[:, None] * inv_freq[None, :]
    ey = torch.stack((sy.sin(), sy.cos()), dim=-1).flatten(-2, -1)     # [h, dim]
    ex = torch.stack((sx.sin(), sx.cos()), dim=-1).flatten(-2, -1)     # [w, dim]
    emb = torch.zeros(h, w, dim * 2)
    emb[:, :, :dim] = ex[None, :, :]
    emb[:, :, dim:] = ey[:, None, :]
    return emb


def get_similarity(mk, ms, qk, qe):
    """Anisotropic L2 similarity.  mk [CK,N], ms [N] or None, qk [CK,M], qe [CK,M] or None -> [N,M].
    cutie/model/utils/memory_utils.py:7-46"""
    CK = mk.shape[0]
    if qe is not None:
        a_sq = mk.pow(2).t() @ qe
        two_ab = 2 * (mk.t() @ (qk * qe))
        b_sq = (qe * qk.pow(2)).sum(0, keepdim=True)
        sim = -a_sq + two_ab - b_sq
    else:
        a_sq = mk.pow(2).sum(0).unsqueeze(1)
        two_ab = 2 * (mk.t() @ qk)
        sim = -a_sq + two_ab
    if ms is not None:
        sim = sim * ms[:, None] / math.sqrt(CK)
    else:
        sim = sim / math.sqrt(CK)
    return sim


def topk_softmax(sim, top_k):
    """Top-k softmax over memory dim 0 (no max shift), dense result + usage row sums.
    cutie/model/utils/memory_utils.py:49-77"""
    if top_k is not None:
        k = min(top_k, sim.shape[0])
        # NOTE: the reference calls topk(k=top_k) and would raise when N < top_k; N >= HW >> 30 always.
        values, indices = torch.topk(sim, k=k, dim=0)
        x_exp = values.exp()
        x_exp = x_exp / x_exp.sum(dim=0, keepdim=True)
        aff = torch.zeros_like(sim).scatter_(0, indices, x_exp)
    else:
        maxes = sim.max(dim=0, keepdim=True)[0]
        x_exp = torch.exp(sim - maxes)
        aff = x_exp / x_exp.sum(dim=0, keepdim=True)
    return aff, aff.sum(dim=1)


class OracleNet:
    def __init__(self, state_dict, m=MODEL_CFG):
        self.W = {k: v.detach().clone().float() if v.is_floating_point() else v.clone()
                  for k, v in state_dict.items()}
        self.m = m
        self.mean = torch.tensor(m['pixel_mean']).view(1, 3, 1, 1)
        self.std = torch.tensor(m['pixel_std']).view(1, 3, 1, 1)
        self._pe_cache = {}

    # ---- primitive layers -------------------------------------------------------
    def conv(self, name, x, stride=1, padding=0):
        return F.conv2d(x, self.W[name + '.weight'], self.W.get(name + '.bias'), stride, padding)

    def bn(self, name, x):
        W = self.W
        return F.batch_norm(x, W[name + '.running_mean'], W[name + '.running_var'],
                            W[name + '.weight'], W[name + '.bias'], False, 0.0, 1e-5)

    def linear(self, name, x):
        return F.linear(x, self.W[name + '.weight'], self.W[name + '.bias'])

    def ln(self, name, x):
        return F.layer_norm(x, (x.shape[-1],), self.W[name + '.weight'], self.W[name + '.bias'], 1e-5)

    def gconv(self, name, g, stride=1, padding=0):
        # group_modules.py:33-37 : conv applied per object
        b, k = g.shape[:2]
        y = self.conv(name, g.flatten(0, 1), stride, padding)
        return y.view(b, k, *y.shape[1:])

    # ---- ResNets ------------------------------------------------------------------
    def _bottleneck(self, p, x, stride):
        # resnet.py:83-124 (stride on the 3x3)
        out = F.relu(self.bn(p + '.bn1', self.conv(p + '.conv1', x)))
        out = F.relu(self.bn(p + '.bn2', self.conv(p + '.conv2', out, stride, 1)))
        out = self.bn(p + '.bn3', self.conv(p + '.conv3', out))
        if (p + '.downsample.0.weight') in self.W:
            x = self.bn(p + '.downsample.1', self.conv(p + '.downsample.0', x, stride))
        return F.relu(out + x)

    def _basic(self, p, x, stride):
        # resnet.py:51-80
        out = F.relu(self.bn(p + '.bn1', self.conv(p + '.conv1', x, stride, 1)))
        out = self.bn(p + '.bn2', self.conv(p + '.conv2', out, 1, 1))
        if (p + '.downsample.0.weight') in self.W:
            x = self.bn(p + '.downsample.1', self.conv(p + '.downsample.0', x, stride))
        return F.relu(out + x)

    def _layer(self, prefix, x, nblocks, stride, block):
        for bi in range(nblocks):
            x = block(f'{prefix}.{bi}', x, stride if bi == 0 else 1)
        return x

    # ---- CUTIE.encode_image (cutie.py:61-64, big_modules.py:45-54) ----------------
    def encode_image(self, image):
        x = (image - self.mean) / self.std
        x = F.relu(self.bn('pixel_encoder.bn1', self.conv('pixel_encoder.conv1', x, 2, 3)))
        x = F.max_pool2d(x, 3, 2, 1)
        if self.m.get('pixel_encoder_type', 'resnet50') == 'resnet18':      # model/small.yaml, big_modules.py:28-29
            block, depth = self._basic, (2, 2, 2)
        else:
            block, depth = self._bottleneck, (3, 4, 6)
        f4 = self._layer('pixel_encoder.res2', x, depth[0], 1, block)
        f8 = self._layer('pixel_encoder.layer2', f4, depth[1], 2, block)
        f16 = self._layer('pixel_encoder.layer3', f8, depth[2], 2, block)
        return (f16, f8, f4), self.conv('pix_feat_proj', f16)

    # ---- CUTIE.transform_key (cutie.py:92-98, big_modules.py:81-87) ---------------
    def transform_key(self, f16):
        x = self.conv('key_proj.pix_feat_proj', f16)
        shrinkage = self.conv('key_proj.d_proj', x, 1, 1) ** 2 + 1
        selection = torch.sigmoid(self.conv('key_proj.e_proj', x, 1, 1))
        key = self.conv('key_proj.key_proj', x, 1, 1)
        return key, shrinkage, selection

    # ---- shared blocks ------------------------------------------------------------
    @staticmethod
    def _others(masks):
        # cutie.py:49-59
        if masks.shape[1] >= 1:
            return (masks.sum(dim=1, keepdim=True) - masks).clamp(0, 1)
        return torch.zeros_like(masks)

    def _ca_block(self, p, x):
        # channel_attn.py:25-39 (in_dim == out_dim, residual)
        r = x
        x = self.conv(p + '.conv1', F.relu(x), 1, 1)
        x = self.conv(p + '.conv2', F.relu(x), 1, 1)
        b, c = x.shape[:2]
        w = x.mean(dim=(2, 3)).view(b, 1, c)
        w = F.conv1d(w, self.W[p + '.conv.weight'], None, 1, 2).transpose(-1, -2).unsqueeze(-1).sigmoid()
        return x * w + r

    def _fusion_block(self, p, x, g):
        # group_modules.py:102-127
        b, k = g.shape[:2]
        g = self.conv(p + '.distributor.x_transform', x).unsqueeze(1) + self.gconv(p + '.distributor.g_transform', g)
        g = g.flatten(0, 1)
        g = self._ca_block(p + '.block1', g)
        g = self._ca_block(p + '.block2', g)
        return g.view(b, k, *g.shape[1:])

    @staticmethod
    def _gru(h, values):
        # modules.py:35-43
        dim = values.shape[2] // 3
        f = torch.sigmoid(values[:, :, :dim])
        u = torch.sigmoid(values[:, :, dim:dim * 2])
        n = torch.tanh(values[:, :, dim * 2:])
        return f * h * (1 - u) + u * n

    def _group_resblock(self, p, g):
        # group_modules.py:40-58
        out = self.gconv(p + '.conv1', F.relu(g), 1, 1)
        out = self.gconv(p + '.conv2', F.relu(out), 1, 1)
        if (p + '.downsample.weight') in self.W:
            g = self.gconv(p + '.downsample', g)
        return out + g

    # ---- CUTIE.encode_mask (cutie.py:66-90, big_modules.py:122-182, object_summarizer.py:55-89)
    def encode_mask(self, image, pix_feat, sensory, masks, deep_update=True):
        img = (image - self.mean) / self.std
        others = self._others(masks)
        k = masks.shape[1]
        g = torch.stack([masks, others], dim=2)                                  # [1,K,2,H,W]
        g = torch.cat([img.unsqueeze(1).expand(-1, k, -1, -1, -1), g], 2)       # [1,K,5,H,W]
        x = g.flatten(0, 1)
        x = self.bn('mask_encoder.bn1', self.conv('mask_encoder.conv1', x, 2, 3))
        x = F.relu(F.max_pool2d(x, 3, 2, 1))                                     # maxpool then relu
        x = self._layer('mask_encoder.layer1', x, 2, 1, self._basic)
        x = self._layer('mask_encoder.layer2', x, 2, 2, self._basic)
        x = self._layer('mask_encoder.layer3', x, 2, 2, self._basic)
        x = x.view(1, k, *x.shape[1:])
        value = self._fusion_block('mask_encoder.fuser', pix_feat, x)
        new_sensory = sensory
        if deep_update:
            # modules.py:71-85
            vals = self.gconv('mask_encoder.sensory_update.transform', torch.cat([value, sensory], 2), 1, 1)
            new_sensory = self._gru(sensory, vals)
        summaries = self.object_summarizer(masks, value)
        return value, new_sensory, summaries

    def _pe(self, h, w):
        key = (h, w)
        if key not in self._pe_cache:
            self._pe_cache[key] = positional_encoding(h, w, self.m['embed_dim'], self.m['pixel_pe_scale'],
                                                      self.m['pixel_pe_temperature'])
        return self._pe_cache[key]

    def object_summarizer(self, masks, value):
        # object_summarizer.py:55-89, _weighted_pooling :11-23
        h, w = value.shape[-2:]
        nq = self.m['num_queries']
        m = F.interpolate(masks, size=(h, w), mode='area').unsqueeze(-1)          # [1,K,h,w,1]
        rep = torch.cat([m.expand(-1, -1, -1, -1, nq // 2), (1 - m).expand(-1, -1, -1, -1, nq // 2)], -1)
        v = value.permute(0, 1, 3, 4, 2)
        v = self.linear('object_summarizer.input_proj', v)
        v = v + self._pe(h, w)[None, None]
        feature = self.linear('object_summarizer.feature_pred.2',
                              F.relu(self.linear('object_summarizer.feature_pred.0', v)))
        logits = self.linear('object_summarizer.weights_pred.2',
                             F.relu(self.linear('object_summarizer.weights_pred.0', v)))
        weights = logits.sigmoid() * rep
        sums = torch.einsum('bkhwq,bkhwc->bkqc', weights, feature)
        area = weights.flatten(2, 3).sum(2).unsqueeze(-1)
        return torch.cat([sums, area], dim=-1)                                    # [1,K,16,257]

    # ---- CUTIE.pixel_fusion (cutie.py:142-157, big_modules.py:207-235) ------------
    def pixel_fusion(self, pix_feat, pixel, sensory, last_mask):
        lm = F.interpolate(last_mask, size=sensory.shape[-2:], mode='area')
        lo = self._others(lm)
        lm2 = torch.stack([lm, lo], dim=2)
        sr = self.gconv('pixel_fuser.sensory_compress', torch.cat([sensory, lm2], 2))
        p16 = pixel + sr
        return self._fusion_block('pixel_fuser.fuser', pix_feat, p16)

    # ---- CUTIE.readout_query (object_transformer.py:114-205, transformer_layers.py) -
    def _mha(self, p, q, k, v, attn_mask=None):
        """nn.MultiheadAttention (batch_first) with separate q/k/v inputs. q [B,L,C], k/v [B,S,C],
        attn_mask bool [B*heads, L, S] True = blocked."""
        C = q.shape[-1]
        nh = self.m['num_heads']
        hd = C // nh
        Wi, bi = self.W[p + '.in_proj_weight'], self.W[p + '.in_proj_bias']
        qp = F.linear(q, Wi[:C], bi[:C])
        kp = F.linear(k, Wi[C:2 * C], bi[C:2 * C])
        vp = F.linear(v, Wi[2 * C:], bi[2 * C:])
        B, L, _ = qp.shape
        S = kp.shape[1]
        qh = qp.view(B, L, nh, hd).transpose(1, 2)
        kh = kp.view(B, S, nh, hd).transpose(1, 2)
        vh = vp.view(B, S, nh, hd).transpose(1, 2)
        att = (qh @ kh.transpose(-1, -2)) / math.sqrt(hd)                         # [B,nh,L,S]
        if attn_mask is not None:
            att = att.masked_fill(attn_mask.view(B, nh, L, S), float('-inf'))
        att = att.softmax(dim=-1)
        out = (att @ vh).transpose(1, 2).reshape(B, L, C)
        return self.linear(p + '.out_proj', out)

    def _aux_mask(self, logits):
        # object_transformer.py:179-205 ; logits [1,K,h,w] -> bool [K*heads, Q, hw], True = blocked
        nh, nq = self.m['num_heads'], self.m['num_queries']
        lg = aggregate(logits.sigmoid(), dim=1)
        is_fg = (lg[:, 1:] >= lg.max(dim=1, keepdim=True)[0])
        fg = is_fg.bool().flatten(2)                                              # [1,K,hw]
        a_fg = (~fg).unsqueeze(2).unsqueeze(2).repeat(1, 1, nh, nq // 2, 1).flatten(0, 2)
        a_bg = fg.unsqueeze(2).unsqueeze(2).repeat(1, 1, nh, nq // 2, 1).flatten(0, 2)
        mask = torch.cat([a_fg, a_bg], dim=1)
        mask[torch.where(mask.sum(-1) == mask.shape[-1])] = False
        return mask

    def readout_query(self, pixel, obj_mem, return_aux=False):
        """pixel [1,K,C,h,w]; obj_mem [1,K,1,Q,C+1] -> pixel' [1,K,C,h,w]."""
        p = 'object_transformer'
        bs, K, C, H, W = pixel.shape
        nq = self.m['num_queries']
        osum = obj_mem.view(bs * K, obj_mem.shape[2], nq, C + 1)
        sums = osum[:, :, :, :-1].sum(dim=1)
        area = osum[:, :, :, -1:].sum(dim=1)
        vals = sums / (area + 1e-4)
        query = self.W[p + '.query_init.weight'].unsqueeze(0) + self.linear(p + '.summary_to_query_init', vals)
        query_emb = self.W[p + '.query_emb.weight'].unsqueeze(0) + self.linear(p + '.summary_to_query_emb', vals)
        pixel_init = self.gconv(p + '.pixel_init_proj', pixel)
        pixel_emb = self.gconv(p + '.pixel_emb_proj', pixel)
        pe = self._pe(H, W).flatten(0, 1)[None]                                   # [1,hw,C]
        pixel_pe = pe + pixel_emb.flatten(3, 4).flatten(0, 1).transpose(1, 2)
        pixel = pixel_init
        aux = []
        lg = self.gconv(p + '.mask_pred.0.1', F.relu(pixel)).squeeze(2)
        aux.append(lg)
        attn_mask = self._aux_mask(lg)
        for b in range(self.m['num_blocks']):
            q = f'{p}.blocks.{b}'
            pixel_flat = pixel.flatten(3, 4).flatten(0, 1).transpose(1, 2)        # [K,hw,C]
            # read_from_pixel: CrossAttention (transformer_layers.py:45-98); residual is the normed x
            x = self.ln(q + '.read_from_pixel.norm', query)
            x = x + self._mha(q + '.read_from_pixel.cross_attn', x + query_emb, pixel_flat + pixel_pe,
                              pixel_flat, attn_mask)
            # self attention (transformer_layers.py:12-41)
            y = self.ln(q + '.self_attn.norm', x)
            x = y + self._mha(q + '.self_attn.self_attn', y + query_emb, y + query_emb, y)
            # FFN (transformer_layers.py:101-118)
            x = x + self.linear(q + '.ffn.linear2', F.relu(self.linear(q + '.ffn.linear1', self.ln(q + '.ffn.norm', x))))
            query = x
            # read_from_query: no norm, residual on pixel
            pixel_flat = pixel_flat + self._mha(q + '.read_from_query.cross_attn', pixel_flat + pixel_pe,
                                                x + query_emb, x)
            # PixelFFN (transformer_layers.py:121-136)
            pf = pixel_flat.view(bs * K, H, W, C).permute(0, 3, 1, 2)
            pixel = self._ca_block(q + '.pixel_ffn.conv', pf).view(bs, K, C, H, W)
            lg = self.gconv(f'{p}.mask_pred.{b + 1}.1', F.relu(pixel)).squeeze(2)
            aux.append(lg)
            attn_mask = self._aux_mask(lg)
        if return_aux:
            return pixel, aux
        return pixel

    # ---- CUTIE.segment (cutie.py:172-203, big_modules.py:257-306, modules.py:8-68) ----
    def segment(self, ms_feats, readout, sensory, update_sensory=True):
        f8 = self.conv('mask_decoder.decoder_feat_proc.transforms.0', ms_feats[1])
        f4 = self.conv('mask_decoder.decoder_feat_proc.transforms.1', ms_feats[2])
        bs, K = readout.shape[:2]

        def up(g):
            y = F.interpolate(g.flatten(0, 1), scale_factor=2, mode='bilinear', align_corners=False)
            return y.view(bs, K, *y.shape[1:])

        def down(g, r):
            y = F.interpolate(g.flatten(0, 1), scale_factor=1 / r, mode='area')
            return y.view(bs, K, *y.shape[1:])

        p16 = readout
        p8 = self._group_resblock('mask_decoder.up_16_8.out_conv', up(p16) + f8.unsqueeze(1))
        p4 = self._group_resblock('mask_decoder.up_8_4.out_conv', up(p8) + f4.unsqueeze(1))
        logits = self.conv('mask_decoder.pred', F.relu(p4.flatten(0, 1)), 1, 1)    # [K,1,h4,w4]
        new_sensory = sensory
        if update_sensory:
            p4c = torch.cat([p4, logits.view(bs, K, 1, *logits.shape[-2:])], 2)
            g = (self.gconv('mask_decoder.sensory_update.g16_conv', p16)
                 + self.gconv('mask_decoder.sensory_update.g8_conv', down(p8, 2))
                 + self.gconv('mask_decoder.sensory_update.g4_conv', down(p4c, 4)))
            vals = self.gconv('mask_decoder.sensory_update.transform', torch.cat([g, sensory], 2), 1, 1)
            new_sensory = self._gru(sensory, vals)
        logits = logits.view(bs, K, *logits.shape[-2:])
        prob = torch.sigmoid(logits)
        lg = aggregate(prob, dim=1)
        lg = F.interpolate(lg, scale_factor=4, mode='bilinear', align_corners=False)
        prob = F.softmax(lg, dim=1)
        return new_sensory, lg, prob
