"""Input side of the hot path: the batch *contract* of txt2vid/data/__init__.py (`Dataset` :158-258,
`collate_fn` :326-355, `my_dataset` :376-377) on synthetic Moving-MNIST-shaped clips (no dataset, no
network, no cv2 / DALI), plus the pinned-memory prefetcher that replaces `data_prefetcher` (:131-156).

A batch is (videos [B,T,C,H,W] float32 in [-1,1], tokens [B,L] int64 zero-padded, lengths list desc).
"""
import threading

import numpy as np
import torch

# vocabulary of the synthetic captions (generate.py:102-182): 4 specials + 17 words = 21
WORDS = ['<pad>', '<start>', '<end>', '<unk>', 'digit', '0', '1', '2', '3', '4', '5', '6', '7', '8', '9', 'is', 'left',
         'and', 'right', 'top', 'bottom']
MOTIONS = [('left', 'right'), ('right', 'left'), ('top', 'bottom'), ('bottom', 'top')]


class Vocab(object):
    """Minimal stand-in for `txt2vid.data.Vocab` (:260-316): word <-> id, `to_words`."""

    def __init__(self, words=WORDS):
        self.words = list(words)
        self.index = {w: i for i, w in enumerate(self.words)}

    def __len__(self):
        return len(self.words)

    def __call__(self, w):
        return self.index.get(w, self.index['<unk>'])

    def to_words(self, ids):
        return ' '.join(self.words[int(i)] for i in ids if int(i) != 0)


class SyntheticMovingDigits(torch.utils.data.Dataset):
    """One 28x28 blob (values U[-1,1]) translating with a bounce between two edge points over a -1
    background, `num_frames` frames of `size` x `size`, `channels` channels; caption
    `<start> digit N is A and B <end>` (8 tokens). Mirrors the statistics of
    txt2vid/data/synthetic/generate.py:18-47,136-170. Deterministic per (seed, index)."""

    def __init__(self, length=1024, num_frames=16, size=64, channels=1, seed=100, vocab=None):
        self.length, self.num_frames, self.size, self.channels, self.seed = length, num_frames, size, channels, seed
        self.vocab = vocab or Vocab()

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        rs = np.random.RandomState((self.seed * 1000003 + i) & 0x7FFFFFFF)
        T, S, C = self.num_frames, self.size, self.channels
        blob = rs.uniform(-1, 1, size=(28, 28)).astype(np.float32)
        digit, mot = int(rs.randint(10)), int(rs.randint(4))
        lim = S - 28
        fixed = int(rs.randint(lim + 1))
        vid = -np.ones((T, C, S, S), dtype=np.float32)
        for t in range(T):
            ph = t / float(T - 1) * 2.0                       # there and back
            pos = int(round((ph if ph <= 1 else 2 - ph) * lim))
            if mot in (1, 3):
                pos = lim - pos
            y, x = (fixed, pos) if mot < 2 else (pos, fixed)
            vid[t, :, y:y + 28, x:x + 28] = blob
        a, b = MOTIONS[mot]
        cap = ['<start>', 'digit', str(digit), 'is', a, 'and', b, '<end>']
        return torch.from_numpy(vid), torch.tensor([self.vocab(w) for w in cap], dtype=torch.float32)


def collate_fn(data):
    """Sort by caption length (desc), stack videos, zero-pad tokens — data/__init__.py:326-355."""
    data = sorted(data, key=lambda d: len(d[1]), reverse=True)
    vids = torch.stack([d[0] for d in data], 0)
    lengths = [len(d[1]) for d in data]
    toks = torch.zeros(len(data), max(lengths), dtype=torch.long)
    for i, d in enumerate(data):
        toks[i, :lengths[i]] = d[1].long()
    return vids, toks, lengths


def my_dataset(data=None, vocab=None, anno=None, transform=None, num_frames=16, **args):
    """Factory named by config/*.json (`"class": "txt2vid.data.my_dataset"`, data/__init__.py:376-377).
    There is no dataset on disk in this build: a `data` path that does not exist (or 'synthetic')
    yields the synthetic clips; reading real frame folders is SURVEY §8(f) item 1 (next)."""
    import os
    if data is not None and data != 'synthetic' and os.path.isdir(str(data)):
        raise NotImplementedError('frame-folder datasets are SURVEY §8(f)-1 (not built yet); use data="synthetic"')
    args.pop('transform', None)
    return SyntheticMovingDigits(num_frames=num_frames, vocab=vocab, **args)


def get_loader(dset=None, batch_size=64, val=False, num_workers=0, has_captions=True):
    """data/__init__.py:379-383."""
    return torch.utils.data.DataLoader(dset, batch_size=batch_size, shuffle=not val, num_workers=num_workers,
                                       collate_fn=collate_fn, drop_last=True, pin_memory=True)


class DevicePrefetcher(object):
    """Overlaps the next batch's host->device copy with the current step: a side HIP stream copies from
    pinned memory; `next()` makes the compute stream wait on it. Same protocol as `data_prefetcher`
    (data/__init__.py:131-156): returns (x, [tokens, lengths]) or (None, None) at the end."""

    def __init__(self, loader, device):
        self.it = iter(loader)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self._preload()

    def _preload(self):
        try:
            batch = next(self.it)
        except StopIteration:
            self.nx, self.ny = None, None
            return
        x, rest = batch[0], list(batch[1:])
        if self.stream is None:
            self.nx, self.ny = x, rest
            return
        with torch.cuda.stream(self.stream):
            self.nx = x.to(self.device, non_blocking=True)
            self.ny = [a.to(self.device, non_blocking=True) if isinstance(a, torch.Tensor) else a for a in rest]

    def next(self):
        if self.nx is None:
            return None, None
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self.nx.record_stream(torch.cuda.current_stream(self.device))
        x, y = self.nx, self.ny
        self._preload()
        return x, y
