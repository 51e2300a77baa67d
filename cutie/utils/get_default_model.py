from cutie_amd.utils.get_default_model import get_default_model  # noqa: F401
