from cutie_amd.inference.memory_manager import MemoryManager  # noqa: F401
