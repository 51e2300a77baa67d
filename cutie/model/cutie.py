from cutie_amd.model.cutie import CUTIE  # noqa: F401
