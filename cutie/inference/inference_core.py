from cutie_amd.inference.inference_core import InferenceCore  # noqa: F401
