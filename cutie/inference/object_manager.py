from cutie_amd.inference.object_manager import ObjectManager  # noqa: F401
