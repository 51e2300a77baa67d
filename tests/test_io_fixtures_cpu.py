"""VideoReader / ResultSaver of the product against behaviour recorded from the EXECUTED reference classes
(oracle/make_io_fixtures.py -> tests/golden/io/: the input folder, what the reference reader returned for nine option
combinations, and the files the reference writer produced for seven calls).  Frames, masks, ids, info fields: exact.  PNG files:
byte-identical (same Pillow writes them from identical arrays); the visualisation JPEGs: decoded pixels identical."""
import base64
import io
import json
import os

import numpy as np
import pytest
from PIL import Image

from oracle.make_io_fixtures import IO, READER_CASES, SAVER_CASES, read_case, run_saver_case

GOLD_READER = np.load(os.path.join(IO, 'video_reader.npz'))
GOLD_SAVER = json.load(open(os.path.join(IO, 'result_saver.json')))


@pytest.mark.parametrize('name', sorted(READER_CASES))
def test_video_reader_matches_the_reference(name):
    from cutie_amd.inference.data.video_reader import VideoReader
    got = read_case(VideoReader, name)
    want = {k[len(name) + 1:]: GOLD_READER[k] for k in GOLD_READER.files if k.startswith(name + '/')}
    assert sorted(got) == sorted(want)
    for k, w in want.items():
        g = np.asarray(got[k])
        assert g.dtype == w.dtype and g.shape == w.shape, (name, k, g.dtype, w.dtype, g.shape, w.shape)
        assert np.array_equal(g, w), (name, k)


@pytest.mark.parametrize('name', sorted(SAVER_CASES))
def test_result_saver_matches_the_reference(name, tmp_path):
    from cutie_amd.inference.data.video_reader import VideoReader
    from cutie_amd.inference.object_manager import ObjectManager
    from cutie_amd.inference.utils.results_utils import ResultSaver
    files = run_saver_case(ResultSaver, ObjectManager, VideoReader, name, str(tmp_path))
    want = GOLD_SAVER[name]
    assert sorted(files) == sorted(want), (name, sorted(files), sorted(want))
    for rel, rec in want.items():
        img = Image.open(io.BytesIO(files[rel]))
        arr = np.array(img)
        assert img.mode == rec['mode'] and list(arr.shape) == rec['shape'] and str(arr.dtype) == rec['dtype'], (name, rel, img.mode, arr.shape)
        assert arr.tobytes() == base64.b64decode(rec['pixels_b64']), (name, rel)
        if rec['palette'] is not None:
            assert img.getpalette() == rec['palette'], (name, rel)
        if rel.endswith('.png'):
            assert files[rel] == base64.b64decode(rec['bytes_b64']), (name, rel, 'PNG bytes')
