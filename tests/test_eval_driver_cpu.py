"""Section 8(f) rank 1 on CPU: VideoReader / VOSTestDataset / ResultSaver / eval driver on a generated video folder, the HIP ops
executed by the torch interpreter of the descriptors (tests/mock_exec.py)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from cutie_amd import _lib
from cutie_amd.config import default_config
from cutie_amd.inference.utils.results_utils import davis_palette, davis_palette_np, voc_palette
from oracle.weights import make_state_dict

from mock_exec import MockExecutor


@pytest.fixture(scope='module')
def product_net():
    from cutie_amd.model.cutie import CUTIE
    mx = MockExecutor()
    mx.per_sample_conv = True       # the driver announces its next frames (a batched encoder plan): the interpreter's convs must round the same at any batch size
    _lib.set_executor_for_testing(mx)
    net = CUTIE(default_config())
    net.load_weights(make_state_dict(seed=0))
    yield net
    _lib.set_executor_for_testing(None)


def _make_video(root, name, n=4, h=64, w=96, ids=(1, 3)):
    from cutie_amd.utils.synth import SyntheticClip
    clip = SyntheticClip(h, w, len(ids), n, seed=9)
    os.makedirs(os.path.join(root, 'JPEGImages', name)); os.makedirs(os.path.join(root, 'Annotations', name))
    for t in range(n):
        arr = (clip.frame(t).permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)
        Image.fromarray(arr).save(os.path.join(root, 'JPEGImages', name, f'{t:05d}.jpg'), quality=95)
    m = clip.first_mask().numpy()
    lut = np.zeros(256, dtype=np.uint8)
    for k, oid in enumerate(ids):
        lut[k + 1] = oid
    png = Image.fromarray(lut[m].astype(np.uint8))
    png.putpalette(davis_palette)
    png.save(os.path.join(root, 'Annotations', name, '00000.png'))
    return lut[m]


def test_palette_is_the_voc_colour_map():
    assert davis_palette_np.shape == (256, 3)
    assert davis_palette_np[:4].tolist() == [[0, 0, 0], [128, 0, 0], [0, 128, 0], [128, 128, 0]]
    assert voc_palette(256)[255].tolist() == [224, 224, 192]


def test_video_reader_semantics(tmp_path):
    from cutie_amd.inference.data.vos_test_dataset import VOSTestDataset
    first = _make_video(str(tmp_path), 'vidA', n=3, h=64, w=96, ids=(1, 3))
    ds = VOSTestDataset(os.path.join(tmp_path, 'JPEGImages'), os.path.join(tmp_path, 'Annotations'), use_all_masks=False, size=48)
    assert len(ds) == 1
    rd = next(iter(ds.get_datasets()))
    assert rd.vid_name == 'vidA' and len(rd) == 3 and not rd.use_long_id and rd.get_palette() is not None
    d0, d1 = rd[0], rd[1]
    assert d0['rgb'].shape == (3, 48, 72) and d0['rgb'].dtype == torch.float32 and 0 <= float(d0['rgb'].min()) and float(d0['rgb'].max()) <= 1
    assert d0['info']['resize_needed'] and d0['info']['shape'] == (64, 96) and d0['info']['frame'] == '00000.jpg'
    assert d0['mask'].shape == (48, 72) and sorted(d0['valid_labels'].tolist()) == [1, 3]
    assert 'mask' not in d1 and d1['info']['time_index'] == 1
    full = VOSTestDataset(os.path.join(tmp_path, 'JPEGImages'), os.path.join(tmp_path, 'Annotations'), use_all_masks=False)
    r0 = next(iter(full.get_datasets()))[0]
    assert not r0['info']['resize_needed'] and torch.equal(r0['mask'], torch.from_numpy(first).long())


def test_eval_driver_writes_palette_pngs(tmp_path, product_net):
    """Whole loop (reader -> InferenceCore.step -> fused argmax/remap -> writer thread): the first PNG is the input mask with
    the original object ids and palette; later frames agree with output_prob_to_mask on the same probabilities."""
    from cutie_amd.eval_vos import process_video
    from cutie_amd.inference.data.vos_test_dataset import VOSTestDataset
    from cutie_amd.inference.inference_core import InferenceCore
    first = _make_video(str(tmp_path), 'vidB', n=3, h=64, w=96, ids=(2, 5))
    ds = VOSTestDataset(os.path.join(tmp_path, 'JPEGImages'), os.path.join(tmp_path, 'Annotations'), use_all_masks=False)
    rd = next(iter(ds.get_datasets()))
    out = os.path.join(tmp_path, 'out')
    with torch.inference_mode():
        r = process_video(product_net, default_config(mem_every=2), rd, out, dataset='d17-val', visualize=True,
                          visualize_output_root=os.path.join(tmp_path, 'vis'))
        assert r['frames'] == 3 and r['seconds'] > 0
        files = sorted(os.listdir(os.path.join(out, 'vidB')))
        assert files == ['00000.png', '00001.png', '00002.png']
        p0 = Image.open(os.path.join(out, 'vidB', '00000.png'))
        assert p0.mode == 'P' and p0.getpalette()[:6] == [0, 0, 0, 128, 0, 0]
        assert np.array_equal(np.array(p0), first)
        assert sorted(os.listdir(os.path.join(tmp_path, 'vis', 'vidB'))) == ['00000.jpg', '00001.jpg', '00002.jpg']
        # reference semantics of the id remap on a second pass
        proc = InferenceCore(product_net, cfg=default_config(mem_every=2))
        d = [rd[t] for t in range(3)]
        proc.step(d[0]['rgb'], d[0]['mask'], d[0]['valid_labels'].tolist())
        prob = proc.step(d[1]['rgb'])
        ids = proc.output_prob_to_mask(prob)
        assert ids.dtype == torch.int64 and set(torch.unique(ids).tolist()) <= {0, 2, 5}
        lut = torch.tensor([0, 2, 5])
        assert torch.equal(ids, lut[prob.argmax(0)])
        assert np.array_equal(np.array(Image.open(os.path.join(out, 'vidB', '00001.png'))), ids.numpy().astype(np.uint8))
        assert torch.equal(proc.output_prob_to_mask(prob, dtype=torch.uint8).long(), ids)


def test_videos_in_lock_step_write_the_same_pngs(tmp_path, product_net):
    """eval_vos.process_videos_lockstep (the --lockstep option of the dataset driver): two videos of one frame size and object count, of
    different lengths, advanced in lock step write byte for byte the PNGs that process_video writes for each of them alone; lockstep_key
    tells which videos may share a group."""
    from cutie_amd.eval_vos import lockstep_key, process_video, process_videos_lockstep
    from cutie_amd.inference.data.vos_test_dataset import VOSTestDataset

    def make(name, n, ids, seed):
        from cutie_amd.utils.synth import SyntheticClip
        root = str(tmp_path)
        clip = SyntheticClip(64, 96, len(ids), n, seed=seed)
        os.makedirs(os.path.join(root, 'JPEGImages', name)); os.makedirs(os.path.join(root, 'Annotations', name))
        for t in range(n):
            arr = (clip.frame(t).permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)
            Image.fromarray(arr).save(os.path.join(root, 'JPEGImages', name, f'{t:05d}.jpg'), quality=95)
        lut = np.zeros(256, dtype=np.uint8)
        for k, oid in enumerate(ids):
            lut[k + 1] = oid
        png = Image.fromarray(lut[clip.first_mask().numpy()].astype(np.uint8))
        png.putpalette(davis_palette)
        png.save(os.path.join(root, 'Annotations', name, '00000.png'))
    make('vA', 5, (1, 2), 21)
    make('vB', 7, (4, 9), 22)
    make('vC', 4, (1, 2, 3), 23)
    ds = VOSTestDataset(os.path.join(tmp_path, 'JPEGImages'), os.path.join(tmp_path, 'Annotations'), use_all_masks=False)
    rds = {rd.vid_name: rd for rd in ds.get_datasets()}
    assert lockstep_key(rds['vA']) == lockstep_key(rds['vB']) == ((64, 96), 2, False) and lockstep_key(rds['vC']) == ((64, 96), 3, False)
    cfg = default_config(mem_every=2)
    with torch.inference_mode():
        alone = {}
        for n in ('vA', 'vB'):
            alone[n] = process_video(product_net, cfg, rds[n], os.path.join(tmp_path, 'alone'))
        st = process_videos_lockstep(product_net, cfg, [rds['vA'], rds['vB']], os.path.join(tmp_path, 'ls'))
    assert st[0]['frames'] == 5 and st[1]['frames'] == 7 and st[0]['seconds'] > 0
    for n, T in (('vA', 5), ('vB', 7)):
        fa, fl = sorted(os.listdir(os.path.join(tmp_path, 'alone', n))), sorted(os.listdir(os.path.join(tmp_path, 'ls', n)))
        assert fa == fl and len(fa) == T
        for f in fa:
            assert open(os.path.join(tmp_path, 'alone', n, f), 'rb').read() == open(os.path.join(tmp_path, 'ls', n, f), 'rb').read(), (n, f)


def test_video_reader_options_and_long_ids(tmp_path):
    """start / end / reverse / enabled_frame_list / to_save / use_all_masks, RGB long-id masks (id = R + 256 G + 65536 B) and the
    long-id writer (random colour per object, as the reference); make_zip layouts."""
    import shutil
    from cutie_amd.inference.data.video_reader import VideoReader
    from cutie_amd.inference.object_manager import ObjectManager
    from cutie_amd.inference.utils.results_utils import ResultSaver, make_zip
    root = str(tmp_path)
    _make_video(root, 'v', n=5, h=32, w=48, ids=(1, 2))
    img_dir, msk_dir = os.path.join(root, 'JPEGImages', 'v'), os.path.join(root, 'Annotations', 'v')
    rd = VideoReader('v', img_dir, msk_dir, start=1, end=4)
    assert [rd[i]['info']['frame'] for i in range(len(rd))] == ['00001.jpg', '00002.jpg', '00003.jpg'] and rd[0]['info']['time_index'] == 1
    rd = VideoReader('v', img_dir, msk_dir, reverse=True, to_save=['00003'])
    assert rd[0]['info']['frame'] == '00004.jpg' and [rd[i]['info']['save'] for i in range(5)] == [False, True, False, False, False]
    rd = VideoReader('v', img_dir, msk_dir, enabled_frame_list=['00000', '00002'])
    assert len(rd) == 2 and 'mask' in rd[0]
    # a second annotated frame is only read with use_all_masks
    shutil.copy(os.path.join(msk_dir, '00000.png'), os.path.join(msk_dir, '00003.png'))
    assert 'mask' not in VideoReader('v', img_dir, msk_dir)[3] and 'mask' in VideoReader('v', img_dir, msk_dir, use_all_masks=True)[3]
    # long ids: RGB annotation -> integer ids -> RGB output
    lmsk = os.path.join(root, 'long', 'v')
    os.makedirs(lmsk)
    ids = np.zeros((32, 48), dtype=np.int64); ids[4:12, 5:20] = 70000; ids[15:30, 10:40] = 300
    rgb = np.stack([ids & 255, (ids >> 8) & 255, (ids >> 16) & 255], -1).astype(np.uint8)
    Image.fromarray(rgb).save(os.path.join(lmsk, '00000.png'))
    rd = VideoReader('v', img_dir, lmsk)
    assert rd.use_long_id and rd.get_palette() is None
    d0 = rd[0]
    assert torch.equal(d0['mask'], torch.from_numpy(ids)) and sorted(d0['valid_labels'].tolist()) == [300, 70000]
    om = ObjectManager()
    om.add_new_objects([300, 70000])
    out = os.path.join(root, 'out')
    saver = ResultSaver(out, 'v', dataset='generic', object_manager=om, use_long_id=True)
    prob = torch.zeros(3, 32, 48); prob[0] = 0.4; prob[1, 15:30, 10:40] = 0.9; prob[2, 4:12, 5:20] = 0.9
    saver.process(prob, '00000.jpg')
    saver.end()
    # the long-id writer gives every object a random colour (reference results_utils.py:171-178 + pano_utils.py; recorded
    # behaviour: tests/test_io_fixtures_cpu.py): one colour per object, background black, the object's pixels exactly
    back = np.array(Image.open(os.path.join(out, 'v', '00000.png'))).astype(np.int64)
    code = back[..., 0] + 256 * back[..., 1] + 65536 * back[..., 2]
    colours = {int(i): set(code[ids == i].tolist()) for i in (0, 300, 70000)}
    assert colours[0] == {0} and all(len(colours[i]) == 1 for i in (300, 70000)) and colours[300] != colours[70000]
    assert all(next(iter(colours[i])) >= 255 for i in (300, 70000))
    with pytest.raises(NotImplementedError):
        ResultSaver(out, 'v', dataset='burst-val', object_manager=om, use_long_id=False)
    # archive layouts
    run = os.path.join(root, 'run'); os.makedirs(os.path.join(run, 'Annotations', 'v'))
    shutil.copy(os.path.join(out, 'v', '00000.png'), os.path.join(run, 'Annotations', 'v'))
    make_zip('y19-val', run, 'exp', os.path.join(run, 'Annotations'))
    make_zip('d17-test-dev', run, 'exp', os.path.join(run, 'Annotations'))
    make_zip('d17-val', run, 'exp', os.path.join(run, 'Annotations'))
    assert os.path.exists(os.path.join(run, 'exp_y19-val.zip')) and os.path.exists(os.path.join(run, 'exp_d17-test-dev.zip'))
    assert not os.path.exists(os.path.join(run, 'exp_d17-val.zip'))


def test_score_dumps_and_multi_scale_merge(tmp_path, product_net):
    """Section 8(f) rank 4: save_scores dumps (uint8 prob x255, backward map on the last frame) of a plain and a flip_aug run,
    merged like scripts/merge_multi_scale.py: float sum -> argmax -> tmp id -> object id -> palette PNG (+ DAVIS zip)."""
    from cutie_amd.eval_vos import process_video
    from cutie_amd.inference.data.vos_test_dataset import VOSTestDataset
    from cutie_amd.merge_multi_scale import load_backward, load_scores, merge
    _make_video(str(tmp_path), 'vidC', n=3, h=64, w=96, ids=(2, 5))
    ds = VOSTestDataset(os.path.join(tmp_path, 'JPEGImages'), os.path.join(tmp_path, 'Annotations'), use_all_masks=False)
    rd = next(iter(ds.get_datasets()))
    runs = []
    with torch.inference_mode():
        for name, flip in (('runA', False), ('runB', True)):
            root = os.path.join(tmp_path, name)
            process_video(product_net, default_config(mem_every=2, flip_aug=flip), rd, os.path.join(root, 'Annotations'),
                          dataset='d17-val', save_scores=True, score_output_root=os.path.join(root, 'Scores'))
            runs.append(root)
    sc = os.path.join(runs[0], 'Scores', 'vidC')
    assert sorted(os.listdir(sc)) == ['00000.npz', '00001.npz', '00002.npz', 'backward.npz']
    assert load_backward(sc) == {2: 1, 5: 2}
    a, b = load_scores(os.path.join(sc, '00001.npz')), load_scores(os.path.join(runs[1], 'Scores', 'vidC', '00001.npz'))
    assert a.dtype == np.uint8 and a.shape == (3, 64, 96) and b.shape == a.shape
    assert np.abs(a.astype(np.int32).sum(0) - 255).max() <= 3            # truncated probabilities of a softmax
    # the dump of a single run reproduces that run's PNG wherever the quantised argmax is unambiguous
    png = np.array(Image.open(os.path.join(runs[0], 'Annotations', 'vidC', '00001.png')))
    srt = np.sort(a.astype(np.int32), 0)
    clear = srt[-1] > srt[-2]
    assert np.array_equal(np.array([0, 2, 5])[a.argmax(0)][clear], png[clear])
    out = os.path.join(tmp_path, 'merged')
    assert merge(runs, out, dataset='D', num_proc=1) == 3
    assert os.path.exists(out + '.zip')
    want = np.array([0, 2, 5], dtype=np.uint8)[(a.astype(np.float32) + b).argmax(0)]
    got = Image.open(os.path.join(out, 'vidC', '00001.png'))
    assert got.mode == 'P' and np.array_equal(np.array(got), want)
    with pytest.raises(ValueError):
        from cutie_amd.inference.utils.results_utils import ResultSaver
        ResultSaver(out, 'x', dataset='d17-val', object_manager=None, use_long_id=False, save_scores=True)


def test_process_video_workflow(tmp_path, product_net):
    """Section 8(f) rank 3: scripts/process_video.py -- masks committed to permanent memory first, then the whole video,
    memory clean-up by device-memory ratio.  (The step sequence itself is pinned to the reference by scenario small_video.)"""
    from cutie_amd.process_video import check_to_clear_non_permanent_memory, one_hot_planes, process_video, video_config
    from cutie_amd.utils.synth import SyntheticClip
    clip = SyntheticClip(64, 96, 2, 7, seed=4)
    frames, masks, out = os.path.join(tmp_path, 'frames'), os.path.join(tmp_path, 'masks'), os.path.join(tmp_path, 'out')
    os.makedirs(frames); os.makedirs(masks)
    for t in range(7):
        Image.fromarray((clip.frame(t).permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)).save(
            os.path.join(frames, f'{t:07d}.png'))
    m0 = clip.first_mask().numpy().astype(np.uint8)
    for t in (0, 4):
        png = Image.fromarray(m0); png.putpalette(davis_palette); png.save(os.path.join(masks, f'{t:07d}.png'))
    assert torch.equal(one_hot_planes(m0, 2, 'cpu'), torch.nn.functional.one_hot(torch.from_numpy(m0).long(), 3).permute(2, 0, 1).float()[1:])
    with pytest.raises(RuntimeError):
        one_hot_planes(m0, 1, 'cpu')
    cfg = video_config(mem_every=2, max_internal_size=-1)
    assert cfg.use_long_term and video_config().mem_every == 10 and video_config().max_internal_size == 480
    calls = []

    def fake_info():                                    # 95 % used on the 4th query only
        calls.append(1)
        return (5, 100) if len(calls) == 4 else (60, 100)

    r = process_video(product_net, cfg, frames, masks, out, mem_cleanup_ratio=0.9, mem_get_info=fake_info)
    assert r['frames'] == 7 and r['num_objects'] == 2 and r['cleanups'] == 1 and len(calls) == 7
    proc = r['processor']
    HW = (64 // 16) * (96 // 16)
    # two committed masks + the same two met again in the video (each first-in-bucket / forced insert is permanent)
    assert proc.memory.work_mem.perm_size(0) >= 2 * HW
    assert sorted(os.listdir(out)) == [f'{t:07d}.png' for t in range(7)]
    p0 = Image.open(os.path.join(out, '0000000.png'))
    assert p0.mode == 'P' and np.array_equal(np.array(p0), m0)          # a one-hot mask comes back as itself
    assert not check_to_clear_non_permanent_memory(proc, -1, fake_info) and len(calls) == 7


def test_read_ahead_is_ordered_and_bounded():
    """The threaded loader yields source[i] in order, keeps at most ``depth`` items in flight, surfaces a reader error at the
    frame it belongs to, and degrades to inline reads with workers=0."""
    import threading, time
    from cutie_amd.inference.data.prefetch import ReadAhead
    started, lock = [], threading.Lock()

    class Src:
        def __len__(self):
            return 23

        def __getitem__(self, i):
            with lock:
                started.append(i)
            time.sleep(0.002 * ((i * 7) % 5))                 # out-of-order completion
            if i == 17:
                raise ValueError('bad frame 17')
            return i * i

    got = []
    with pytest.raises(ValueError, match='bad frame 17'):
        for k, v in enumerate(ReadAhead(Src(), workers=4, depth=5)):
            got.append(v)
            with lock:
                assert max(started) <= k + 5                  # never more than `depth` frames ahead of the consumer
    assert got == [i * i for i in range(17)]
    assert list(ReadAhead(list(range(9)), workers=0)) == list(range(9))
    assert list(ReadAhead(None, workers=2, length=4, getitem=lambda i: -i)) == [0, -1, -2, -3]
    assert list(ReadAhead([], workers=3)) == []
