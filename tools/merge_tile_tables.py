"""Merge conv tile tables (the format of cutie_amd/tiles_gfx950.json / conv_sweep's *_tiles.json): later files override earlier ones.
    python tools/merge_tile_tables.py out.json base.json sweep_a_tiles.json sweep_b_tiles.json ..."""
import json
import sys

out, srcs = sys.argv[1], sys.argv[2:]
table = {}
for f in srcs:
    for k, v in json.load(open(f))['tiles']:
        table[tuple(k)] = list(v)
json.dump({'tiles': [[list(k), v] for k, v in table.items()]}, open(out, 'w'))
print(f'{len(table)} geometries -> {out}')
