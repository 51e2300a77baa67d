"""Parameter inventory of the reference ``CUTIE`` module (names, shapes, roles).

``CUTIE.state_dict()`` of the reference has 527 tensors; checkpoints such as
``cutie-base-mega.pth`` are flat dicts with exactly these keys, so the drop-in module must expose the same
names.  Roles: 'w' conv/linear weight, 'b' bias, 'bn_*' BatchNorm pieces, 'emb' embedding, 'ln_w'/'ln_b'
LayerNorm, 'eca' the k=5 Conv1d of CAResBlock, 'buf' a persistent non-trainable buffer (inv_freq).

References: cutie/model/cutie.py:18-47, big_modules.py, modules.py, group_modules.py, channel_attn.py,
transformer/*.py, utils/resnet.py:127-166, aux_modules.py:40-60.
"""
from collections import OrderedDict


class _Spec(OrderedDict):
    def conv(self, name, cout, cin, k, bias=True):
        self[name + '.weight'] = ((cout, cin, k, k), 'w')
        if bias:
            self[name + '.bias'] = ((cout,), 'b')

    def lin(self, name, cout, cin):
        self[name + '.weight'] = ((cout, cin), 'w')
        self[name + '.bias'] = ((cout,), 'b')

    def bn(self, name, c):
        for suffix, role in (('weight', 'bn_w'), ('bias', 'bn_b'), ('running_mean', 'bn_mean'), ('running_var', 'bn_var')):
            self[f'{name}.{suffix}'] = ((c,), role)
        self[name + '.num_batches_tracked'] = ((), 'bn_nbt')

    def ln(self, name, c):
        self[name + '.weight'] = ((c,), 'ln_w')
        self[name + '.bias'] = ((c,), 'ln_b')

    def mha(self, name, c):
        self[name + '.in_proj_weight'] = ((3 * c, c), 'w')
        self[name + '.in_proj_bias'] = ((3 * c,), 'b')
        self.lin(name + '.out_proj', c, c)

    def ca_block(self, name, c):
        self.conv(name + '.conv1', c, c, 3)
        self.conv(name + '.conv2', c, c, 3)
        self[name + '.conv.weight'] = ((1, 1, 5), 'eca')

    def fusion(self, name, cx, cg, cout):
        self.conv(name + '.distributor.x_transform', cout, cx, 1)
        self.conv(name + '.distributor.g_transform', cout, cg, 1)
        self.ca_block(name + '.block1', cout)
        self.ca_block(name + '.block2', cout)

    def resnet_trunk(self, prefix, bottleneck, in_ch, first_layer):
        self.conv(prefix + '.conv1', 64, in_ch, 7, bias=False)
        self.bn(prefix + '.bn1', 64)
        depth = (3, 4, 6) if bottleneck else (2, 2, 2)
        exp = 4 if bottleneck else 1
        cin = 64
        for li, nb in enumerate(depth):
            planes = 64 << li
            lname = first_layer if li == 0 else f'layer{li + 1}'
            for bi in range(nb):
                p = f'{prefix}.{lname}.{bi}'
                if bottleneck:
                    shapes = [(planes, cin, 1), (planes, planes, 3), (planes * 4, planes, 1)]
                else:
                    shapes = [(planes, cin, 3), (planes, planes, 3)]
                for ci, (co, cc, k) in enumerate(shapes, 1):
                    self.conv(f'{p}.conv{ci}', co, cc, k, bias=False)
                    self.bn(f'{p}.bn{ci}', co)
                if bi == 0 and (li > 0 or cin != planes * exp):
                    self.conv(p + '.downsample.0', planes * exp, cin, 1, bias=False)
                    self.bn(p + '.downsample.1', planes * exp)
                cin = planes * exp


def resnet_layers(m, prefix):
    """-> (bottleneck?, [(layer attr, planes, blocks, stride)]) of a trunk (resnet.py:127-179; the pixel encoder calls its
    first stage ``res2``, big_modules.py:39).  model/base.yaml: ResNet-50 pixel encoder; model/small.yaml: ResNet-18."""
    first = 'res2' if prefix == 'pixel_encoder' else 'layer1'
    kind = m[prefix]['type']
    assert kind in ('resnet18', 'resnet50'), kind
    depth = (3, 4, 6) if kind == 'resnet50' else (2, 2, 2)
    return kind == 'resnet50', [(first, 64, depth[0], 1), ('layer2', 128, depth[1], 2), ('layer3', 256, depth[2], 2)]



def build_spec(m):
    s = _Spec()
    C, CK, CV, CS, CE = m['pixel_dim'], m['key_dim'], m['value_dim'], m['sensory_dim'], m['embed_dim']
    ms = m['pixel_encoder']['ms_dims']
    up = m['mask_decoder']['up_dims']
    ot = m['object_transformer']
    assert m['mask_encoder']['type'] == 'resnet18', 'the mask encoder of every released model is a ResNet-18'
    s.resnet_trunk('pixel_encoder', resnet_layers(m, 'pixel_encoder')[0], 3, 'res2')
    assert ms[0] == (1024 if m['pixel_encoder']['type'] == 'resnet50' else 256), 'ms_dims do not match the pixel encoder'
    s.conv('pix_feat_proj', C, ms[0], 1)
    s.conv('key_proj.pix_feat_proj', C, ms[0], 1)
    s.conv('key_proj.key_proj', CK, C, 3)
    s.conv('key_proj.d_proj', 1, C, 3)
    s.conv('key_proj.e_proj', CK, C, 3)
    s.resnet_trunk('mask_encoder', False, 5, 'layer1')
    s.fusion('mask_encoder.fuser', C, m['mask_encoder']['final_dim'], CV)
    s.conv('mask_encoder.sensory_update.transform', CS * 3, CV + CS, 3)
    s.conv('mask_decoder.sensory_update.g16_conv', CS, up[0], 1)
    s.conv('mask_decoder.sensory_update.g8_conv', CS, up[1], 1)
    s.conv('mask_decoder.sensory_update.g4_conv', CS, up[2] + 1, 1)
    s.conv('mask_decoder.sensory_update.transform', CS * 3, CS * 2, 3)
    s.conv('mask_decoder.decoder_feat_proc.transforms.0', up[0], ms[1], 1)
    s.conv('mask_decoder.decoder_feat_proc.transforms.1', up[1], ms[2], 1)
    s.conv('mask_decoder.up_16_8.out_conv.downsample', up[1], up[0], 1)
    s.conv('mask_decoder.up_16_8.out_conv.conv1', up[1], up[0], 3)
    s.conv('mask_decoder.up_16_8.out_conv.conv2', up[1], up[1], 3)
    s.conv('mask_decoder.up_8_4.out_conv.conv1', up[2], up[1], 3)
    s.conv('mask_decoder.up_8_4.out_conv.conv2', up[2], up[2], 3)
    s.conv('mask_decoder.pred', 1, up[2], 3)
    s.fusion('pixel_fuser.fuser', C, CV, CE)
    s.conv('pixel_fuser.sensory_compress', CV, CS + 2, 1)
    t = 'object_transformer'
    nq = ot['num_queries']
    s[t + '.query_init.weight'] = ((nq, CE), 'emb')
    s[t + '.query_emb.weight'] = ((nq, CE), 'emb')
    s.lin(t + '.summary_to_query_init', CE, CE)
    s.lin(t + '.summary_to_query_emb', CE, CE)
    s.conv(t + '.pixel_init_proj', CE, CE, 1)
    s.conv(t + '.pixel_emb_proj', CE, CE, 1)
    s[t + '.spatial_pe.inv_freq'] = ((CE // 4,), 'buf')
    for b in range(ot['num_blocks']):
        q = f'{t}.blocks.{b}'
        s.mha(q + '.read_from_pixel.cross_attn', CE)
        s.ln(q + '.read_from_pixel.norm', CE)
        s.mha(q + '.self_attn.self_attn', CE)
        s.ln(q + '.self_attn.norm', CE)
        s.lin(q + '.ffn.linear1', ot['ff_dim'], CE)
        s.lin(q + '.ffn.linear2', CE, ot['ff_dim'])
        s.ln(q + '.ffn.norm', CE)
        s.mha(q + '.read_from_query.cross_attn', CE)
        s.ca_block(q + '.pixel_ffn.conv', CE)
    for b in range(ot['num_blocks'] + 1):
        s.conv(f'{t}.mask_pred.{b}.1', 1, CE, 1)
    u = 'object_summarizer'
    s[u + '.pos_enc.inv_freq'] = ((CE // 4,), 'buf')
    s.lin(u + '.input_proj', CE, CV)
    s.lin(u + '.feature_pred.0', CE, CE)
    s.lin(u + '.feature_pred.2', CE, CE)
    s.lin(u + '.weights_pred.0', CE, CE)
    s.lin(u + '.weights_pred.2', nq, CE)
    s.conv('aux_computer.sensory_aux.projection', CE + 1, CS, 1)      # training-only head; kept so checkpoints load
    return s
