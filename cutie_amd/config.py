"""Hydra-free configuration for the inference hot path.

The reference composes ``cutie/config/eval_config.yaml`` + ``model/base.yaml`` with hydra/omegaconf and then
hoists per-dataset values (``cutie/inference/utils/args_utils.py:7-30``).  Only the hyper-parameters the hot
path reads are kept here; ``Config`` supports the four access styles the reference uses on its cfg
(``cfg.x``, ``cfg['x']``, ``cfg.get('x')``, ``'x' in cfg``).
"""
import copy


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def wrap(o):
    if isinstance(o, dict):
        return Config({k: wrap(v) for k, v in o.items()})
    return o


# cutie/config/model/base.yaml:1-58 (interpolations resolved)
MODEL_BASE = dict(
    pixel_mean=[0.485, 0.456, 0.406], pixel_std=[0.229, 0.224, 0.225],
    pixel_dim=256, key_dim=64, value_dim=256, sensory_dim=256, embed_dim=256,
    pixel_encoder=dict(type='resnet50', ms_dims=[1024, 512, 256]),
    mask_encoder=dict(type='resnet18', final_dim=256),
    pixel_pe_scale=32, pixel_pe_temperature=128,
    object_transformer=dict(embed_dim=256, ff_dim=2048, num_heads=8, num_blocks=3, num_queries=16,
                            read_from_pixel=dict(input_norm=False, input_add_pe=False, add_pe_to_qkv=[True, True, False]),
                            read_from_query=dict(add_pe_to_qkv=[True, True, False], output_norm=False),
                            query_self_attention=dict(add_pe_to_qkv=[True, True, False])),
    object_summarizer=dict(embed_dim=256, num_summaries=16, add_pe=True),
    aux_loss=dict(sensory=dict(enabled=True, weight=0.01), query=dict(enabled=True, weight=0.01)),
    mask_decoder=dict(up_dims=[256, 128, 128]),
)

# cutie/config/model/small.yaml: base with a ResNet-18 pixel encoder
MODEL_SMALL = dict(MODEL_BASE, pixel_encoder=dict(type='resnet18', ms_dims=[256, 128, 64]))
MODELS = {'base': MODEL_BASE, 'small': MODEL_SMALL}

# cutie/config/eval_config.yaml:11-51 + datasets.d17-val (use_long_term False, mem_every 5) hoisted
EVAL_DEFAULTS = dict(
    exp_id='default', dataset='d17-val', amp=False, weights='output/cutie-base-mega.pth', flip_aug=False,
    max_internal_size=-1, use_long_term=False, mem_every=5, max_mem_frames=5,
    long_term=dict(count_usage=True, max_mem_frames=10, min_mem_frames=5, num_prototypes=128,
                   max_num_tokens=10000, buffer_tokens=2000),
    top_k=30, stagger_updates=5, chunk_size=-1, save_scores=False, save_aux=False, visualize=False,
)


def default_config(**overrides):
    """eval_config defaults; ``model='small'`` selects model/small.yaml like hydra's ``model=small`` override."""
    cfg = copy.deepcopy(EVAL_DEFAULTS)
    cfg['model'] = copy.deepcopy(MODEL_BASE)
    for k, v in overrides.items():
        cfg[k] = copy.deepcopy(MODELS[v]) if (k == 'model' and isinstance(v, str)) else copy.deepcopy(v)
    return wrap(cfg)
