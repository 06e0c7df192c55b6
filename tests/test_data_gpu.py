"""GPU: the input row in front of the hot path — synthetic Moving-MNIST-shaped clips generated in HBM equal the
host dataset's items bit for bit (integer / geometry work), and the reference-pinned frame-folder pipeline
(tests/golden/data_contract.npz) reaches HBM unchanged through Dataset -> collate_fn -> DataLoader -> DevicePrefetcher."""
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('size,channels,frames,seed', [(64, 1, 16, 100), (128, 3, 16, 7), (32, 2, 5, 3), (28, 1, 2, 11)])
def test_device_synthetic_clips_equal_host_items(size, channels, frames, seed):
    """`t2v_synth_clips` vs `SyntheticMovingDigits.__getitem__`: numpy's RandomState (MT19937 seeding / twist / tempering, 53-bit
    random_sample, masked-rejection randint), the bounce trajectory with Python's round-half-even and the caption ids are
    reproduced on the device: every float of every frame and every token equal (`torch.equal`)."""
    from txt2vid_amd import data as D
    ds = D.SyntheticMovingDigits(length=1 << 20, num_frames=frames, size=size, channels=channels, seed=seed)
    idx = [0, 1, 2, 3, 17, 255, 1023, 99999, (1 << 20) - 1]
    vids, toks, lengths = ds.device_batch(idx, DEV)
    torch.cuda.synchronize()
    assert int(ds._err.item()) == 0 and lengths == [8] * len(idx)
    assert vids.shape == (len(idx), frames, channels, size, size) and toks.dtype == torch.int64
    motions = set()
    for k, i in enumerate(idx):
        v, c = ds[i]
        assert torch.equal(vids[k].cpu(), v), (i, float((vids[k].cpu() - v).abs().max()))
        assert torch.equal(toks[k].cpu(), c.long()), i
        motions.add(tuple(c.long().tolist()[4:7]))
    assert len(motions) >= 2 and (size == 28 or float(vids.min()) == -1.0) and float(vids.max()) <= 1.0
    # into caller-owned buffers, device-resident indices, and what collate_fn makes of the host items
    out = (torch.empty_like(vids), torch.empty_like(toks))
    ds.device_batch(torch.tensor(idx, device=DEV), DEV, out=out)
    hv, ht, hl = D.collate_fn([ds[i] for i in idx])
    assert torch.equal(out[0].cpu(), hv) and torch.equal(out[1].cpu(), ht) and hl == lengths


def test_device_synthetic_captions_follow_the_vocabulary():
    """A vocabulary built in another word order (`build_vocab` over a sentence list) gives other ids: the kernel takes them from the host."""
    from txt2vid_amd import data as D
    vocab = D.build_vocab(['bottom top right and left is 9 8 7 6 5 4 3 2 1 0 digit.'])
    ds = D.SyntheticMovingDigits(length=64, size=64, seed=5, vocab=vocab)
    assert vocab('digit') != D.Vocab()('digit')
    _, toks, _ = ds.device_batch(range(16), DEV)
    for k in range(16):
        assert torch.equal(toks[k].cpu(), ds[k][1].long())
    with pytest.raises(RuntimeError):
        ds.device_batch(range(4), 'cpu')                      # no CPU fallback: index the dataset instead


def test_device_loader_feeds_the_training_loop_shapes():
    """`get_loader(on_device dataset, device=cuda)` -> DeviceSyntheticLoader -> DevicePrefetcher -> channel-first clip: the same
    batches, in the same order, as the host DataLoader path from the same generator state."""
    from txt2vid_amd import data as D
    from txt2vid_amd import functional as TF
    ds_dev = D.my_dataset(data='synthetic', vocab=None, size=64, channels=1, seed=9, length=24, on_device=True)
    ds_host = D.my_dataset(data='synthetic', vocab=None, size=64, channels=1, seed=9, length=24)
    torch.manual_seed(42)
    host = list(D.get_loader(dset=ds_host, batch_size=8, num_workers=0, device=DEV))
    torch.manual_seed(42)
    pre = D.DevicePrefetcher(D.get_loader(dset=ds_dev, batch_size=8, device=DEV), DEV)
    n = 0
    x, y = pre.next()
    while x is not None:
        hv, ht, hl = host[n]
        assert x.is_cuda and torch.equal(x.cpu(), hv) and torch.equal(y[0].cpu(), ht) and list(y[1]) == list(hl)
        xc = TF.video_to_channel_first(x)
        assert torch.equal(xc.cpu(), hv.permute(0, 2, 1, 3, 4).contiguous())
        n += 1
        x, y = pre.next()
    assert n == 3


def test_reference_pinned_frame_folders_reach_hbm_unchanged(golden, tmp_path):
    """tests/golden/data_contract.npz (what the REAL reference's data module produced for two JPEG frame folders) through this
    build's Dataset -> collate_fn -> DataLoader (pinned) -> DevicePrefetcher: the batch that lands in HBM equals the fixture's
    `collate_fn` output — videos, zero-padded tokens, lengths — and the channel-first view the loop consumes is its permutation."""
    from txt2vid_amd import data as D
    from txt2vid_amd import functional as TF
    g = golden('data_contract')
    vocab = D.build_vocab([str(s) for s in g['sentences']])
    for key in g.files:
        if key.startswith('jpeg_'):
            _, vid, idx = key.split('_')
            (tmp_path / vid).mkdir(exist_ok=True)
            (tmp_path / vid / ('%s.jpg' % idx)).write_bytes(g[key].tobytes())
    cap_path = tmp_path / 'captions.pkl'
    cap_path.write_bytes(pickle.dumps(pickle.loads(g['captions_pickle'].tobytes())))

    def transform(img):
        a = np.asarray(img.convert('L'), dtype=np.float32) / 255.0
        return torch.from_numpy((a[None] - 0.5) / 0.5)
    ds = D.my_dataset(data=str(tmp_path), vocab=vocab, anno=str(cap_path), transform=transform)
    loader = D.get_loader(dset=ds, batch_size=len(ds), val=True, num_workers=0, device=DEV)
    pre = D.DevicePrefetcher(loader, DEV)
    x, y = pre.next()
    torch.cuda.synchronize()
    assert x.is_cuda and y[0].is_cuda
    assert np.array_equal(x.cpu().numpy(), g['collate_vids'])
    assert np.array_equal(y[0].cpu().numpy(), g['collate_targets']) and str(y[0].dtype) == str(g['collate_targets_dtype'])
    assert list(y[1]) == list(g['collate_lengths'])
    xc = TF.video_to_channel_first(x)
    assert np.array_equal(xc.cpu().numpy(), np.ascontiguousarray(g['collate_vids'].transpose(0, 2, 1, 3, 4)))
    assert pre.next() == (None, None)
