from cutie_amd.inference.object_info import ObjectInfo  # noqa: F401
