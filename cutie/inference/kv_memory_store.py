# The reference's KeyValueMemoryStore is only ever constructed by its MemoryManager; what callers touch is the gauge surface of
# `processor.memory.work_mem` / `.long_mem` (size / perm_size / non_perm_size / engaged / num_objects), served here by the
# read-only view over the HBM-resident bank.
from cutie_amd.inference.kv_memory_store import StoreView as KeyValueMemoryStore  # noqa: F401
