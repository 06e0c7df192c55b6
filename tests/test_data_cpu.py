"""CPU: the input side of the hot path — frame-folder `Dataset`, `Vocab`, `collate_fn`, `default_transform`,
the `my_dataset` factory and rank sharding — against the contract of txt2vid/data/__init__.py:158-383
(restated in the docstrings; the reference module itself hard-imports cv2 + DALI and cannot be imported)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch
from PIL import Image

from txt2vid_amd import data
from txt2vid_amd.util.reflection import create_object


@pytest.fixture
def frame_tree(tmp_path):
    """3 videos x 40 numbered frames of 80x72 RGB (frame n of video v is the constant colour (n, 10v, 255-n)); one is
    stored as .jpg to cover both suffixes; captions pickle has 2 sentences for video 0 and names a missing video."""
    root = tmp_path / 'videos'
    for v in range(3):
        d = root / ('vid%d' % v)
        d.mkdir(parents=True)
        for n in range(40):
            img = Image.fromarray(np.full((72, 80, 3), (n, 10 * v, 255 - n), dtype=np.uint8))
            img.save(d / ('%d.%s' % (n, 'jpg' if v == 2 else 'png')), **({'quality': 100} if v == 2 else {}))
        (d / 'notes.txt').write_text('ignored')
    caps = {'vid0': ['digit 3 is left and right.', 'a Digit moves'], 'vid1': ['digit 7 is top and bottom.'],
            'vid2': ['digit 1 is right and left.'], 'gone': ['never read']}
    anno = tmp_path / 'sent.pickle'
    with open(anno, 'wb') as f:
        pickle.dump(caps, f)
    return str(root), str(anno)


def test_vocab_contract():
    v = data.Vocab()
    assert len(v) == 21 and [v(w) for w in ('<pad>', '<start>', '<end>', '<unk>')] == [0, 1, 2, 3]
    assert v('DIGIT') == v('digit') and v('zebra') == 3
    assert list(v.tokenize('digit 3 is left.')) == ['<start>', 'digit', '3', 'is', 'left', '<end>']
    assert v.to_words([1, v('digit'), v('3'), 2]) == '<start> digit 3<end>'
    assert v.get_word(10 ** 6) == '<unk>'
    b = data.build_vocab(['Hello world.', 'hello there'])
    assert len(b) == 4 + 3 and b('world') != 3 and b('world.') == 3


def test_frame_folder_dataset_items(frame_tree):
    root, anno = frame_tree
    vocab = data.Vocab()
    ds = data.my_dataset(data=root, vocab=vocab, anno=anno, transform=data.default_transform([64], 1), num_frames=16)
    assert isinstance(ds, data.Dataset) and len(ds) == 4 and ds.missing == 1          # one item per (video, sentence)
    frames, toks = ds[0]
    assert frames.shape == (16, 1, 64, 64) and frames.dtype == torch.float32
    # strided pick: 40 // 16 = 2 -> frames 0, 2, ..., 30, numerically (not lexically) ordered
    n = np.arange(16) * 2
    grey = np.floor((n * 299 + 0 * 587 + (255 - n) * 114 + 500) / 1000.0)                 # PIL's ITU-R 601-2 luma
    want = (grey / 255.0 - 0.5) / 0.5
    got = frames[:, 0, 0, 0].numpy()
    assert np.abs(got - want).max() <= 1.0 / 255 + 1e-6
    assert float(frames.min()) >= -1 and float(frames.max()) <= 1
    assert toks.tolist() == [float(vocab(w)) for w in ['<start>', 'digit', '3', 'is', 'left', 'and', 'right', '<end>']]
    _, toks1 = ds[1]                                       # no trailing '.', unknown words: <end> appended, <unk> ids
    assert toks1.tolist() == [1.0, 3.0, float(vocab('digit')), 3.0, 2.0]
    rgb = data.my_dataset(data=root, vocab=vocab, anno=anno, transform=data.default_transform([48, 56], 3))
    f3, _ = rgb[3]                                         # the .jpg video
    assert f3.shape == (16, 3, 48, 56)
    assert abs(float(f3[1, 1, 0, 0]) - (20 / 255.0 - 0.5) / 0.5) < 0.03


def test_random_frames_are_sorted_and_distinct():
    np.random.seed(0)
    ids = list(range(100, 140))
    got = data.pick_frames(ids, num_frames=8, random=True)
    assert len(got) == 8 and got == sorted(set(got)) and set(got) <= set(ids)


def test_collate_and_loader_contract(frame_tree):
    root, anno = frame_tree
    ds = data.my_dataset(data=root, vocab=data.Vocab(), anno=anno, transform=data.default_transform([64], 1))
    loader = data.get_loader(dset=ds, batch_size=4, val=True, num_workers=0)
    vids, toks, lengths = next(iter(loader))
    assert vids.shape == (4, 16, 1, 64, 64) and toks.dtype == torch.int64
    assert lengths == sorted(lengths, reverse=True) == [8, 8, 8, 5]
    assert toks.shape == (4, 8) and toks[3, 5:].tolist() == [0, 0, 0]


def test_reference_style_json_spec_and_synthetic_fallback(frame_tree, tmp_path):
    root, anno = frame_tree
    spec = tmp_path / 'cfg.json'
    spec.write_text(json.dumps({'class': 'txt2vid.data.my_dataset', 'args': {'data': root, 'num_frames': 16}}))
    ds = create_object(str(spec), vocab=data.Vocab(), anno=anno, transform=data.default_transform([64], 1))
    assert isinstance(ds, data.Dataset) and len(ds) == 4
    spec.write_text(json.dumps({'class': 'txt2vid.data.my_dataset',
                                'args': {'data': '/run/media/doubleu/Linux/synthetic/train/videos', 'num_frames': 16}}))
    syn = create_object(str(spec), vocab=data.Vocab(), anno=None, transform=None, size=64, channels=1, seed=3)
    assert isinstance(syn, data.SyntheticMovingDigits) and syn[0][0].shape == (16, 1, 64, 64)


def test_rank_sharding_is_disjoint_and_covers_the_epoch(frame_tree):
    root, anno = frame_tree
    ds = data.my_dataset(data=root, vocab=data.Vocab(), anno=anno, transform=data.default_transform([64], 1))
    seen = []
    for rank in range(2):
        loader = data.get_loader(dset=ds, batch_size=1, num_workers=0, rank=rank, world=2, seed=5)
        loader.sampler.set_epoch(0)
        seen.append(sorted(loader.sampler))
    assert not set(seen[0]) & set(seen[1]) and sorted(seen[0] + seen[1]) == [0, 1, 2, 3]


def test_vocab_pickles_written_by_the_reference_resolve(tmp_path):
    from txt2vid_amd.util.pick import load
    v = data.build_vocab(['digit 3 is left.'])
    blob = pickle.dumps(v, protocol=0).replace(b'txt2vid_amd.data', b'txt2vid.data')   # the name a reference-side pickle holds
    p = tmp_path / 'vocab.pickle'
    p.write_bytes(blob)
    w = load(str(p))
    assert isinstance(w, data.Vocab) and w('left') == v('left') and len(w) == len(v)


def test_input_pipeline_vs_reference_fixture(golden, tmp_path):
    """tests/golden/data_contract.npz holds what the REAL reference's `txt2vid.data` produced (make_golden.py `data`: Vocab /
    build_vocab / tokenize / to_words, pick_frames, Dataset.__getitem__ on two frame folders whose JPEG bytes are in the
    fixture, collate_fn). This build's data module must reproduce all of it exactly (closes SURVEY §8 f1's pin)."""
    import pickle
    import numpy as np
    import torch
    from txt2vid_amd import data as D
    g = golden('data_contract')
    sentences = [str(s) for s in g['sentences']]
    vocab = D.build_vocab(sentences)
    assert [vocab.idx2word[i] for i in range(len(vocab))] == [str(w) for w in g['vocab_words']]
    probes = [str(s) for s in g['probes']]
    toks = [[vocab(t) for t in vocab.tokenize(s)] for s in probes]
    assert [len(t) for t in toks] == list(g['probe_lens'])
    assert sum(toks, []) == list(g['probe_tokens'])
    assert [vocab.to_words(t) for t in toks] == [str(w) for w in g['probe_words']]
    for n in (16, 17, 40, 64):
        assert D.pick_frames(list(range(100, 100 + n)), num_frames=16, random=False) == list(g['pick_%d' % n])
    # the two frame folders, byte for byte
    for key in g.files:
        if key.startswith('jpeg_'):
            _, vid, idx = key.split('_')
            (tmp_path / vid).mkdir(exist_ok=True)
            (tmp_path / vid / ('%s.jpg' % idx)).write_bytes(g[key].tobytes())
    for vid in ('vidA', 'vidB'):
        (tmp_path / vid / 'notes.txt').write_text('x')
    captions = pickle.loads(g['captions_pickle'].tobytes())
    cap_path = tmp_path / 'captions.pkl'
    cap_path.write_bytes(pickle.dumps(captions))

    def transform(img):
        a = np.asarray(img.convert('L'), dtype=np.float32) / 255.0
        return torch.from_numpy((a[None] - 0.5) / 0.5)
    ds = D.Dataset(video_dir=str(tmp_path), vocab=vocab, captions=str(cap_path), transform=transform)
    assert len(ds) == int(g['ds_len']) and ds.missing == int(g['ds_missing'])
    assert [str(v) for v in ds.video_ids] == [str(v) for v in g['ds_video_ids']]
    items = [ds[i] for i in range(len(ds))]
    for i, (frames, cap) in enumerate(items):
        assert np.array_equal(frames.numpy(), g['ds_frames_%d' % i]), i
        assert np.array_equal(cap.numpy(), g['ds_caption_%d' % i]) and str(cap.dtype) == str(g['ds_caption_dtype_%d' % i])
    vids, targets, lengths = D.collate_fn(list(items))
    assert np.array_equal(vids.numpy(), g['collate_vids'])
    assert np.array_equal(targets.numpy(), g['collate_targets']) and str(targets.dtype) == str(g['collate_targets_dtype'])
    assert list(lengths) == list(g['collate_lengths'])
    # the same through the config factory (config/*.json: "class": "txt2vid.data.my_dataset")
    ds2 = D.my_dataset(data=str(tmp_path), vocab=vocab, anno=str(cap_path), transform=transform)
    assert np.array_equal(ds2[1][0].numpy(), g['ds_frames_1'])


def test_device_loader_walks_the_clips_in_the_dataloader_order():
    """`DeviceSyntheticLoader` (clips generated in HBM) visits the samples in the order torch's DataLoader(shuffle=True) does from the
    same global generator state — host and device input rows are interchangeable mid-experiment (order only: no GPU needed)."""
    import torch
    ds = data.SyntheticMovingDigits(length=37, size=32, num_frames=4)

    class Idx(torch.utils.data.Dataset):
        def __len__(self):
            return 37

        def __getitem__(self, i):
            return i
    for shuffle in (True, False):
        torch.manual_seed(123)
        host = [int(i) for b in torch.utils.data.DataLoader(Idx(), batch_size=5, shuffle=shuffle, drop_last=True) for i in b]
        after_host = torch.get_rng_state()
        torch.manual_seed(123)
        loader = data.DeviceSyntheticLoader(ds, 5, 'cuda:0', shuffle=shuffle)
        order = loader.epoch_order()
        assert [int(i) for i in order[:len(loader) * 5]] == host and len(loader) == 7
        assert torch.equal(torch.get_rng_state(), after_host)          # the global generator advanced identically
    # the factory hands the device loader out only for on_device synthetic clips and a CUDA device
    on = data.my_dataset(data='synthetic', vocab=None, size=32, channels=1, seed=3, on_device=True)
    assert isinstance(data.get_loader(dset=on, batch_size=4, device='cuda:0'), data.DeviceSyntheticLoader)
    assert isinstance(data.get_loader(dset=on, batch_size=4, device='cpu'), torch.utils.data.DataLoader)
    assert isinstance(data.get_loader(dset=ds, batch_size=4, device='cuda:0'), torch.utils.data.DataLoader)
