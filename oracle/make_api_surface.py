"""Record the Python surface of the reference's hot-path classes by INTROSPECTING THE UNMODIFIED
REFERENCE: for each class the public methods with their parameters (name, kind, default) and properties.  Output:
tests/golden/api_surface.json, which tests/test_api_surface_cpu.py holds the product classes to (a drop-in must accept every
call the reference accepts).  Run in the build container only:

    python oracle/make_api_surface.py

TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
from oracle.make_golden import GOLDEN, import_reference          # noqa: E402


def describe(cls):
    out = {'methods': {}, 'properties': []}
    for name, member in inspect.getmembers(cls):
        if name.startswith('_') and name != '__init__' and name != '__len__':
            continue
        if isinstance(inspect.getattr_static(cls, name), property):
            out['properties'].append(name)
            continue
        if not inspect.isfunction(member):
            continue
        if name not in cls.__dict__:                                # inherited (nn.Module, object): not the class's own surface
            continue
        params = []
        for p in inspect.signature(member).parameters.values():
            if p.name == 'self':
                continue
            d = None if p.default is inspect.Parameter.empty else repr(p.default)
            params.append({'name': p.name, 'kind': p.kind.name, 'default': d})
        out['methods'][name] = params
    return out


def describe_source(file, cls_name=None, fn_name=None):
    """Same record from the SOURCE (ast) for modules whose imports are not installable here (torchvision, pycocotools,
    hydra): a class's public methods, or one module-level function."""
    import ast
    tree = ast.parse(open(file).read())

    def params_of(fn, drop_self):
        a = fn.args
        pos = a.posonlyargs + a.args
        defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
        out = []
        for arg, d in zip(pos, defaults):
            out.append({'name': arg.arg, 'kind': 'POSITIONAL_OR_KEYWORD', 'default': None if d is None else repr(ast.literal_eval(d))})
        if a.vararg:
            out.append({'name': a.vararg.arg, 'kind': 'VAR_POSITIONAL', 'default': None})
        for arg, d in zip(a.kwonlyargs, a.kw_defaults):
            out.append({'name': arg.arg, 'kind': 'KEYWORD_ONLY', 'default': None if d is None else repr(ast.literal_eval(d))})
        if a.kwarg:
            out.append({'name': a.kwarg.arg, 'kind': 'VAR_KEYWORD', 'default': None})
        return out[1:] if drop_self else out

    for node in tree.body:
        if cls_name and isinstance(node, ast.ClassDef) and node.name == cls_name:
            out = {'methods': {}, 'properties': []}
            for f in node.body:
                if not isinstance(f, ast.FunctionDef) or (f.name.startswith('_') and f.name not in ('__init__', '__len__', '__getitem__')):
                    continue
                if any(isinstance(d, ast.Name) and d.id == 'property' for d in f.decorator_list):
                    out['properties'].append(f.name)
                else:
                    out['methods'][f.name] = params_of(f, True)
            return out
        if fn_name and isinstance(node, ast.FunctionDef) and node.name == fn_name:
            return {'methods': {'__call__': params_of(node, False)}, 'properties': []}
    raise KeyError((file, cls_name, fn_name))


def main():
    import_reference()
    from cutie.inference.image_feature_store import ImageFeatureStore
    from cutie.inference.inference_core import InferenceCore
    from cutie.inference.kv_memory_store import KeyValueMemoryStore
    from cutie.inference.memory_manager import MemoryManager
    from cutie.inference.object_info import ObjectInfo
    from cutie.inference.object_manager import ObjectManager
    from cutie.model.cutie import CUTIE
    surface = {
        'cutie.inference.inference_core.InferenceCore': describe(InferenceCore),
        'cutie.inference.image_feature_store.ImageFeatureStore': describe(ImageFeatureStore),
        'cutie.inference.object_manager.ObjectManager': describe(ObjectManager),
        'cutie.inference.object_info.ObjectInfo': describe(ObjectInfo),
        'cutie.inference.memory_manager.MemoryManager': describe(MemoryManager),
        'cutie.inference.kv_memory_store.KeyValueMemoryStore': describe(KeyValueMemoryStore),
        'cutie.model.cutie.CUTIE': describe(CUTIE),
    }
    R = '/root/reference/cutie/'
    surface['cutie.inference.utils.results_utils.ResultSaver'] = describe_source(R + 'inference/utils/results_utils.py', 'ResultSaver')
    surface['cutie.inference.utils.results_utils.make_zip'] = describe_source(R + 'inference/utils/results_utils.py', fn_name='make_zip')
    surface['cutie.inference.data.video_reader.VideoReader'] = describe_source(R + 'inference/data/video_reader.py', 'VideoReader')
    surface['cutie.inference.data.vos_test_dataset.VOSTestDataset'] = describe_source(R + 'inference/data/vos_test_dataset.py', 'VOSTestDataset')
    surface['cutie.utils.get_default_model.get_default_model'] = describe_source(R + 'utils/get_default_model.py', fn_name='get_default_model')
    fn = os.path.join(GOLDEN, 'api_surface.json')
    with open(fn, 'w') as f:
        json.dump(surface, f, indent=1, sort_keys=True)
    for k, v in surface.items():
        print(k, len(v['methods']), 'methods', v['properties'])


if __name__ == '__main__':
    main()
