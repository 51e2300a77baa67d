"""Deterministic, torch-version-independent synthetic weights of the CUTIE architecture (benchmarks, parity tests, the oracle).

No trained checkpoint exists offline, so benchmarks and parity runs use a
synthetic ``state_dict`` with the reference's exact key set and shapes
(reference: ``CUTIE.state_dict()``, cutie/model/cutie.py:18-47; 527 tensors).
Values come from numpy's PCG64 seeded per tensor name, so the same dict is
reproduced bit-for-bit on any box without shipping 140 MB of weights.

Scales are chosen so activations stay O(1) through the residual stacks and every
parameter (BN statistics, biases, ECA kernels ...) is non-trivial, i.e. a wrong
fold or a dropped bias shows up in the parity tests.
"""
from collections import OrderedDict
import zlib
import numpy as np
import torch

MODEL_CFG = dict(
    pixel_mean=[0.485, 0.456, 0.406], pixel_std=[0.229, 0.224, 0.225],
    pixel_dim=256, key_dim=64, value_dim=256, sensory_dim=256, embed_dim=256,
    ms_dims=[1024, 512, 256], up_dims=[256, 128, 128],
    num_heads=8, num_blocks=3, num_queries=16, ff_dim=2048,
    pixel_pe_scale=32, pixel_pe_temperature=128,
    pixel_encoder_type='resnet50',
)
# cutie/config/model/small.yaml: ResNet-18 pixel encoder (the "cutie-small" checkpoints)
MODEL_CFG_SMALL = dict(MODEL_CFG, pixel_encoder_type='resnet18', ms_dims=[256, 128, 64])


def _bn(spec, name, c):
    spec[name + '.weight'] = ((c,), 'bn_w')
    spec[name + '.bias'] = ((c,), 'bn_b')
    spec[name + '.running_mean'] = ((c,), 'bn_mean')
    spec[name + '.running_var'] = ((c,), 'bn_var')
    spec[name + '.num_batches_tracked'] = ((), 'bn_nbt')


def _conv(spec, name, cout, cin, k, bias=True):
    spec[name + '.weight'] = ((cout, cin, k, k), 'conv_w')
    if bias:
        spec[name + '.bias'] = ((cout,), 'bias')


def _linear(spec, name, cout, cin):
    spec[name + '.weight'] = ((cout, cin), 'linear_w')
    spec[name + '.bias'] = ((cout,), 'bias')


def _ln(spec, name, c):
    spec[name + '.weight'] = ((c,), 'ln_w')
    spec[name + '.bias'] = ((c,), 'ln_b')


def _resnet(spec, prefix, kind, in_ch, layer1_name):
    """ResNet-50 (bottleneck, [3,4,6]) or ResNet-18 (basic, [2,2,2]) up to layer3.
    reference: cutie/model/utils/resnet.py:51-166"""
    _conv(spec, prefix + '.conv1', 64, in_ch, 7, bias=False)
    _bn(spec, prefix + '.bn1', 64)
    inplanes = 64
    if kind == 'resnet50':
        blocks, expansion = [3, 4, 6], 4
    else:
        blocks, expansion = [2, 2, 2], 1
    for li, (planes, nb) in enumerate(zip([64, 128, 256], blocks)):
        lname = layer1_name if li == 0 else f'layer{li + 1}'
        stride = 1 if li == 0 else 2
        for bi in range(nb):
            p = f'{prefix}.{lname}.{bi}'
            if kind == 'resnet50':
                _conv(spec, p + '.conv1', planes, inplanes, 1, bias=False)
                _bn(spec, p + '.bn1', planes)
                _conv(spec, p + '.conv2', planes, planes, 3, bias=False)
                _bn(spec, p + '.bn2', planes)
                _conv(spec, p + '.conv3', planes * 4, planes, 1, bias=False)
                _bn(spec, p + '.bn3', planes * 4)
            else:
                _conv(spec, p + '.conv1', planes, inplanes, 3, bias=False)
                _bn(spec, p + '.bn1', planes)
                _conv(spec, p + '.conv2', planes, planes, 3, bias=False)
                _bn(spec, p + '.bn2', planes)
            if bi == 0 and (stride != 1 or inplanes != planes * expansion):
                _conv(spec, p + '.downsample.0', planes * expansion, inplanes, 1, bias=False)
                _bn(spec, p + '.downsample.1', planes * expansion)
            inplanes = planes * expansion


def _ca_block(spec, name, c):
    _conv(spec, name + '.conv1', c, c, 3)
    _conv(spec, name + '.conv2', c, c, 3)
    spec[name + '.conv.weight'] = ((1, 1, 5), 'conv1d_w')


def _fusion_block(spec, name, x_in, g_in, out):
    _conv(spec, name + '.distributor.x_transform', out, x_in, 1)
    _conv(spec, name + '.distributor.g_transform', out, g_in, 1)
    _ca_block(spec, name + '.block1', out)
    _ca_block(spec, name + '.block2', out)


def _mha(spec, name, c):
    spec[name + '.in_proj_weight'] = ((3 * c, c), 'linear_w')
    spec[name + '.in_proj_bias'] = ((3 * c,), 'bias')
    _linear(spec, name + '.out_proj', c, c)


def param_spec(m=MODEL_CFG):
    """name -> (shape, kind) for every tensor of the reference ``CUTIE.state_dict()``."""
    s = OrderedDict()
    C, CK, CV, CS, CE = m['pixel_dim'], m['key_dim'], m['value_dim'], m['sensory_dim'], m['embed_dim']
    ms, up = m['ms_dims'], m['up_dims']
    _resnet(s, 'pixel_encoder', m.get('pixel_encoder_type', 'resnet50'), 3, 'res2')   # big_modules.py:21-54
    _conv(s, 'pix_feat_proj', C, ms[0], 1)                      # cutie.py:36
    _conv(s, 'key_proj.pix_feat_proj', C, ms[0], 1)             # big_modules.py:64-87
    _conv(s, 'key_proj.key_proj', CK, C, 3)
    _conv(s, 'key_proj.d_proj', 1, C, 3)
    _conv(s, 'key_proj.e_proj', CK, C, 3)
    _resnet(s, 'mask_encoder', 'resnet18', 5, 'layer1')         # big_modules.py:90-120
    _fusion_block(s, 'mask_encoder.fuser', C, 256, CV)
    _conv(s, 'mask_encoder.sensory_update.transform', CS * 3, CV + CS, 3)
    # big_modules.py:238-255, modules.py:46-56
    _conv(s, 'mask_decoder.sensory_update.g16_conv', CS, up[0], 1)
    _conv(s, 'mask_decoder.sensory_update.g8_conv', CS, up[1], 1)
    _conv(s, 'mask_decoder.sensory_update.g4_conv', CS, up[2] + 1, 1)
    _conv(s, 'mask_decoder.sensory_update.transform', CS * 3, CS + CS, 3)
    _conv(s, 'mask_decoder.decoder_feat_proc.transforms.0', up[0], ms[1], 1)
    _conv(s, 'mask_decoder.decoder_feat_proc.transforms.1', up[1], ms[2], 1)
    _conv(s, 'mask_decoder.up_16_8.out_conv.downsample', up[1], up[0], 1)
    _conv(s, 'mask_decoder.up_16_8.out_conv.conv1', up[1], up[0], 3)
    _conv(s, 'mask_decoder.up_16_8.out_conv.conv2', up[1], up[1], 3)
    _conv(s, 'mask_decoder.up_8_4.out_conv.conv1', up[2], up[1], 3)
    _conv(s, 'mask_decoder.up_8_4.out_conv.conv2', up[2], up[2], 3)
    _conv(s, 'mask_decoder.pred', 1, up[2], 3)
    _fusion_block(s, 'pixel_fuser.fuser', C, CV, CE)            # big_modules.py:192-205
    _conv(s, 'pixel_fuser.sensory_compress', CV, CS + 2, 1)
    # object_transformer.py:76-112, 12-34; transformer_layers.py
    p = 'object_transformer'
    nq = m['num_queries']
    s[p + '.query_init.weight'] = ((nq, CE), 'emb')
    s[p + '.query_emb.weight'] = ((nq, CE), 'emb')
    _linear(s, p + '.summary_to_query_init', CE, CE)
    _linear(s, p + '.summary_to_query_emb', CE, CE)
    _conv(s, p + '.pixel_init_proj', CE, CE, 1)
    _conv(s, p + '.pixel_emb_proj', CE, CE, 1)
    s[p + '.spatial_pe.inv_freq'] = ((CE // 4,), 'inv_freq')
    for b in range(m['num_blocks']):
        q = f'{p}.blocks.{b}'
        _mha(s, q + '.read_from_pixel.cross_attn', CE)
        _ln(s, q + '.read_from_pixel.norm', CE)
        _mha(s, q + '.self_attn.self_attn', CE)
        _ln(s, q + '.self_attn.norm', CE)
        _linear(s, q + '.ffn.linear1', m['ff_dim'], CE)
        _linear(s, q + '.ffn.linear2', CE, m['ff_dim'])
        _ln(s, q + '.ffn.norm', CE)
        _mha(s, q + '.read_from_query.cross_attn', CE)
        _ca_block(s, q + '.pixel_ffn.conv', CE)
    for b in range(m['num_blocks'] + 1):
        _conv(s, f'{p}.mask_pred.{b}.1', 1, CE, 1)
    p = 'object_summarizer'                                     # object_summarizer.py:26-53
    s[p + '.pos_enc.inv_freq'] = ((CE // 4,), 'inv_freq')
    _linear(s, p + '.input_proj', CE, CV)
    _linear(s, p + '.feature_pred.0', CE, CE)
    _linear(s, p + '.feature_pred.2', CE, CE)
    _linear(s, p + '.weights_pred.0', CE, CE)
    _linear(s, p + '.weights_pred.2', nq, CE)
    _conv(s, 'aux_computer.sensory_aux.projection', CE + 1, CS, 1)   # aux_modules.py:13-16 (training only)
    return s


# (pattern, gain) -- first match wins.  Tuned (oracle/… tune script in DESIGN.md "synthetic weights") so
# that activations stay O(1), similarities land in [-20, 0], logits in [-6, 6] and the frame-to-frame
# recurrence is contractive: parity over a whole trajectory is then a meaningful test.
CONV_GAINS = [
    ('key_proj.key_proj', 2.8), ('key_proj.d_proj', 3.0), ('key_proj.e_proj', 3.0),
    ('mask_decoder.pred', 3.0), ('.mask_pred.', 1.0),
    ('', 1.0),
]
LINEAR_GAINS = [('read_from_query.cross_attn.out_proj', 0.25), ('in_proj_weight', 0.7), ('', 1.0)]


def _conv_gain(name):
    return next(g for p, g in CONV_GAINS if p in name)


def _linear_gain(name):
    return next(g for p, g in LINEAR_GAINS if p in name)


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))


def make_state_dict(seed=0, m=MODEL_CFG):
    """Deterministic fp32 state_dict (torch CPU tensors) for the whole network."""
    sd = OrderedDict()
    for name, (shape, kind) in param_spec(m).items():
        r = _rng(name, seed)
        if kind == 'conv_w':
            fan_in = shape[1] * shape[2] * shape[3]
            v = r.standard_normal(shape) * (_conv_gain(name) / np.sqrt(fan_in))
        elif kind == 'linear_w':
            v = r.standard_normal(shape) * (_linear_gain(name) / np.sqrt(shape[1]))
        elif kind == 'conv1d_w':
            v = r.standard_normal(shape) * 0.6
        elif kind == 'emb':
            v = r.standard_normal(shape) * 0.5
        elif kind == 'bias':
            v = r.standard_normal(shape) * 0.05
            if name == 'mask_decoder.pred.bias':
                v = v - 0.6            # keeps fg/bg balanced with random features
        elif kind == 'bn_w':
            basic = 'mask_encoder' in name or m.get('pixel_encoder_type', 'resnet50') == 'resnet18'
            last = name.endswith('bn3.weight') or (name.endswith('bn2.weight') and basic)
            v = r.uniform(0.7, 1.1, shape) * (0.45 if last else 1.0)
        elif kind == 'bn_b':
            v = r.standard_normal(shape) * 0.05
        elif kind == 'bn_mean':
            v = r.standard_normal(shape) * 0.1
        elif kind == 'bn_var':
            v = r.uniform(0.6, 1.4, shape)
        elif kind == 'bn_nbt':
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        elif kind == 'ln_w':
            v = r.uniform(0.8, 1.2, shape)
        elif kind == 'ln_b':
            v = r.standard_normal(shape) * 0.05
        elif kind == 'inv_freq':
            # positional_encoding.py:29-31 : dim = ceil(256/4)*2 = 128
            dim = int(np.ceil(m['embed_dim'] / 4) * 2)
            t = torch.arange(0, dim, 2).float() / dim
            sd[name] = 1.0 / (m['pixel_pe_temperature'] ** t)
            continue
        else:
            raise KeyError(kind)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd
