"""Per-object record of the ObjectManager.  Surface of cutie/inference/object_info.py:1-24: ``id``, ``poke_count`` with
``poke()`` / ``unpoke()`` (consecutive missed detections, used by the GUI / BURST tooling to retire objects), and identity BY
ID -- an ObjectInfo hashes and compares like its integer id, which is what lets plain ints index ``obj_to_tmp_id``."""
from dataclasses import dataclass, field


@dataclass(eq=False, repr=False)
class ObjectInfo:
    id: int
    poke_count: int = field(default=0, compare=False)

    def poke(self) -> None:
        self.poke_count = self.poke_count + 1

    def unpoke(self) -> None:
        self.poke_count = 0

    def _key(self) -> int:
        return self.id

    def __hash__(self) -> int:                       # == hash(int id): dict lookups with ints land in the same slot
        return hash(self._key())

    def __eq__(self, other) -> bool:
        other_key = other._key() if isinstance(other, ObjectInfo) else other
        return self._key() == other_key

    def __repr__(self) -> str:
        return '(ID: %d)' % self.id
