"""Drop-in import alias: ``cutie.*`` -> the MI355X-native implementation in ``cutie_amd``.

Lets callers written against hkchengrex/Cutie (``scripting_demo.py``: ``from cutie.inference.inference_core import
InferenceCore``; ``from cutie.utils.get_default_model import get_default_model``) run unchanged with this repository
first on ``sys.path``.  Only the inference hot path is provided."""
