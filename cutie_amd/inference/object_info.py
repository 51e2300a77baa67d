"""Per-object metadata; mirrors cutie/inference/object_info.py:1-24 (hash/eq by id so ints can index dicts)."""


class ObjectInfo:
    def __init__(self, id: int):
        self.id = id
        self.poke_count = 0          # number of consecutive missed detections (GUI / BURST tooling)

    def poke(self) -> None:
        self.poke_count += 1

    def unpoke(self) -> None:
        self.poke_count = 0

    def __hash__(self):
        return hash(self.id)

    def __eq__(self, other):
        return self.id == (other if isinstance(other, int) else other.id)

    def __repr__(self):
        return f'(ID: {self.id})'
