"""Input side of the hot path: the batch *contract* of txt2vid/data/__init__.py (`Dataset` :158-258,
`Vocab` :260-316, `collate_fn` :326-355, `default_transform` :357-370, `my_dataset` :376-377): the
frame-folder reader (PIL only: no cv2 / DALI / torchvision / lmdb), synthetic Moving-MNIST-shaped clips for
machines without a dataset, and the pinned-memory prefetcher that replaces `data_prefetcher` (:131-156).

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
    """Word <-> id table with the interface of `txt2vid.data.Vocab` (:260-316): ids 0..3 are
    <pad> <start> <end> <unk>, words are lower-cased, unknown words map to <unk>. Built with no arguments it
    holds the synthetic captions' 21 words (the reference builds it from a sentence pickle, `build_vocab`)."""
    START, END, UNKNOWN, PAD = '<start>', '<end>', '<unk>', '<pad>'

    def __init__(self, words=WORDS):
        self.word2idx, self.idx2word, self.idx = {}, {}, 0
        for w in (self.PAD, self.START, self.END, self.UNKNOWN):
            self.add_word(w)
        for w in words or ():
            self.add_word(w)

    def add_word(self, word):
        word = word.lower()
        if word not in self.word2idx:
            self.word2idx[word] = self.idx
            self.idx2word[self.idx] = word
            self.idx += 1

    def get_word(self, idx):
        return self.idx2word.get(idx, self.UNKNOWN)

    def __call__(self, word):
        return self.word2idx.get(word.lower(), self.word2idx[self.UNKNOWN])

    def __len__(self):
        return len(self.word2idx)

    def tokenize(self, sentence):
        """<start>, then the whitespace-split words; a word ending in '.' yields the word and <end>."""
        yield self.START
        for word in sentence.split():
            if word.endswith('.'):
                yield word[:-1]
                yield self.END
            else:
                yield word

    def to_words(self, tokens):
        out = []
        for i, tok in enumerate(tokens):
            word = self.get_word(int(tok))
            out.append(word if (i == 0 or word == self.END) else ' ' + word)
        return ''.join(out)


def build_vocab(sentences):
    """data/__init__.py:318-324."""
    vocab = Vocab(words=None)
    for sent in sentences:
        for word in vocab.tokenize(sent):
            vocab.add_word(word)
    return vocab


def pick_frames(frame_ids, num_frames=16, random=False, rng=None):
    """Evenly strided pick (`frame_ids[(len // num_frames) * i]`), or `num_frames` random ids kept in
    order — data/__init__.py:108-129 (whose random branch cannot run as written; this is its intent)."""
    if not random:
        step = len(frame_ids) // num_frames
        return [frame_ids[step * i] for i in range(num_frames)]
    assert len(frame_ids) >= num_frames
    rng = rng or np.random
    keep = np.sort(rng.permutation(len(frame_ids))[:num_frames])
    return [frame_ids[i] for i in keep]


def default_transform(frame_size=None, num_channels=3):
    """PIL image -> float32 [C,h,w] in [-1,1]: centre crop, (grey), /255, (x-0.5)/0.5 — the torchvision
    pipeline of data/__init__.py:357-370 without torchvision (absent here)."""
    fs = list(frame_size) if isinstance(frame_size, (list, tuple)) else [frame_size]
    if len(fs) == 1:
        fs.append(fs[0])
    ch, cw = int(fs[0]), int(fs[1])

    def transform(img):
        w, h = img.size
        left, top = int(round((w - cw) / 2.0)), int(round((h - ch) / 2.0))
        img = img.crop((left, top, left + cw, top + ch))            # PIL zero-fills outside, like CenterCrop's pad
        img = img.convert('RGB' if num_channels == 3 else 'L')
        a = np.asarray(img, dtype=np.float32) / 255.0
        a = a[None] if a.ndim == 2 else a.transpose(2, 0, 1)
        return torch.from_numpy(np.ascontiguousarray((a - 0.5) / 0.5))
    return transform


class Dataset(torch.utils.data.Dataset):
    """Frame-folder reader with the contract of `txt2vid.data.Dataset` (:158-258), minus cv2 / DALI / lmdb:
    `captions` is a pickle {video id -> [sentence, ...]}; `video_dir/<id>/<n>.jpg|png` are the frames
    (n integer, sorted numerically). One item per (video, sentence): (frames [T,C,h,w], token ids float)."""

    def __init__(self, video_dir=None, vocab=None, captions=None, transform=None, random_frames=0, num_frames=16,
                 use_lmdb=False):
        if use_lmdb:
            raise NotImplementedError('the caffe2-LMDB cache is not part of this build (no lmdb here)')
        import os
        from ..util.pick import load
        self.video_dir, self.vocab, self.transform = str(video_dir), vocab, transform
        self.random_frames, self.num_frames = int(random_frames or 0), num_frames
        table = load(captions) if isinstance(captions, (str, os.PathLike)) else captions
        self.video_ids, self.captions, self.missing = [], [], 0
        for vid in table:
            if not os.path.isdir(os.path.join(self.video_dir, str(vid))):
                self.missing += 1
                continue
            for cap in table[vid]:
                self.video_ids.append(vid)
                self.captions.append(cap)

    def __len__(self):
        return len(self.captions)

    def __getitem__(self, idx):
        import os
        from PIL import Image
        folder = os.path.join(self.video_dir, str(self.video_ids[idx]))
        ext = {}
        for name in os.listdir(folder):
            stem, suffix = os.path.splitext(name)
            if suffix in ('.jpg', '.png') and stem.isdigit():
                ext[int(stem)] = name
        ids = sorted(ext)
        n = self.random_frames if self.random_frames else self.num_frames
        picked = pick_frames(ids, num_frames=n, random=bool(self.random_frames))
        frames = []
        for i in picked:
            with Image.open(os.path.join(folder, ext[i])) as img:
                img.load()
                frames.append(self.transform(img) if self.transform else img)
        frames = torch.stack(frames)
        toks = [self.vocab(t) for t in self.vocab.tokenize(self.captions[idx])]
        if toks[-1] != self.vocab(self.vocab.END):
            toks.append(self.vocab(self.vocab.END))
        return frames, torch.tensor(toks, dtype=torch.float32)


class SyntheticMovingDigits(torch.utils.data.Dataset):
    """One 28x28 blob (values U[-1,1]) translating with a bounce between two edge points over a -1
    background, `num_frames` frames of `size` x `size`, `channels` channels; caption
    `<start> digit N is A and B <end>` (8 tokens). Mirrors the statistics of
    txt2vid/data/synthetic/generate.py:18-47,136-170. Deterministic per (seed, index)."""

    def __init__(self, length=1024, num_frames=16, size=64, channels=1, seed=100, vocab=None, on_device=False):
        self.length, self.num_frames, self.size, self.channels, self.seed = length, num_frames, size, channels, seed
        self.vocab = vocab or Vocab()
        self.on_device = bool(on_device)       # `get_loader(..., device=cuda)` then generates the batches in HBM (device_batch)

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

    def device_batch(self, indices, device, out=None):
        """The clips `[self[i] for i in indices]` generated STRAIGHT IN HBM by one kernel launch (`t2v_synth_clips`: numpy's
        RandomState reproduced on the device, so every clip and caption equals the host item bit for bit) and already collated:
        (videos [B,T,C,S,S] float32, tokens [B,8] int64, lengths [8]*B). No host pixel ever exists; nothing crosses PCIe but the
        B indices. `indices`: int64 device tensor, or any host sequence; `out`: optional (videos, tokens) buffers to fill."""
        import ctypes as C
        from .._lib import lib, check
        from .._ops import _stream
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise RuntimeError('device_batch generates on the GPU: the HIP path has no CPU fallback (index the dataset instead)')
        idx = indices if isinstance(indices, torch.Tensor) else torch.as_tensor(list(indices), dtype=torch.int64)
        if not idx.is_cuda and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= len(self)):       # (host indices: checked for free)
            raise IndexError('device_batch: indices must lie in [0, %d)' % len(self))
        idx = idx.to(device=dev, dtype=torch.int64).contiguous()
        B = int(idx.numel())
        T, S, Cc = self.num_frames, self.size, self.channels
        if out is None:
            vids = torch.empty((B, T, Cc, S, S), device=dev, dtype=torch.float32)
            toks = torch.empty((B, 8), device=dev, dtype=torch.int64)
        else:
            vids, toks = out
            if tuple(vids.shape) != (B, T, Cc, S, S) or tuple(toks.shape) != (B, 8) or not vids.is_contiguous() or not toks.is_contiguous() \
                    or vids.dtype != torch.float32 or toks.dtype != torch.int64:
                raise ValueError('device_batch: output buffers must be dense [B,T,C,S,S] float32 / [B,8] int64')
        words = ['<start>', 'digit'] + [str(d) for d in range(10)] + ['is', 'left', 'and', 'right', 'top', 'bottom', '<end>']
        ids = (C.c_int32 * len(words))(*[int(self.vocab(w)) for w in words])
        if getattr(self, '_err', None) is None or self._err.device != dev:
            self._err = torch.zeros((1,), dtype=torch.int32).to(dev)            # (zero by copy: no fill kernel)
        check(lib().t2v_synth_clips(C.c_void_p(idx.data_ptr()), B, int(self.seed), T, Cc, S, ids, C.c_void_p(vids.data_ptr()),
                                    C.c_void_p(toks.data_ptr()), C.c_void_p(self._err.data_ptr()), _stream()), 't2v_synth_clips')
        return vids, toks, [8] * B

    def check_device_draws(self):
        """Raise if any clip generated by `device_batch` since the last check ran out of its draw sequence (the kernel flags it
        in a device word and falls back to digit / motion / position 0: the clip would silently differ from the host item).
        One 4-byte read-back: called once per epoch by `DeviceSyntheticLoader`, and by whoever builds a batch pool."""
        err = getattr(self, '_err', None)
        if err is not None and int(err.item()) != 0:
            err.zero_()
            raise RuntimeError('t2v_synth_clips: a clip exhausted its random draws (device flag set): device batches are not '
                               'bit-identical to the host dataset')


class DeviceSyntheticLoader(object):
    """Stands in for `DataLoader(SyntheticMovingDigits, ...)` when the clips are generated on the GPU: same iteration protocol
    (len, iter -> (videos, tokens, lengths), drop_last), same sample order — a shuffling epoch draws its permutation exactly as
    torch's RandomSampler does (a 64-bit seed from the global torch generator, then `randperm` on a private generator), so host
    and device loaders started from the same generator state walk the same clips. Batches come out on the device."""

    def __init__(self, dset, batch_size, device, shuffle=True):
        self.dataset, self.batch_size, self.device, self.shuffle = dset, int(batch_size), torch.device(device), shuffle
        self.sampler = None

    def __len__(self):
        return len(self.dataset) // self.batch_size

    def epoch_order(self):
        """The sample order of one epoch, consuming the global torch generator exactly as `iter(DataLoader(shuffle=True))` + its
        first `next()` do: the iterator's base seed first, then the RandomSampler's seed for `randperm` on a private generator."""
        n = len(self.dataset)
        torch.empty((), dtype=torch.int64).random_()                 # (_BaseDataLoaderIter's base seed: drawn, unused here)
        if not self.shuffle:
            return torch.arange(n)
        gen = torch.Generator()
        gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        return torch.randperm(n, generator=gen)

    def __iter__(self):
        order = self.epoch_order()
        for k in range(len(self)):
            yield self.dataset.device_batch(order[k * self.batch_size:(k + 1) * self.batch_size], self.device)
        self.dataset.check_device_draws()                            # (one read-back per epoch: flagged, not hidden)


def collate_fn(data):
    """Sort by caption length (desc), stack videos, zero-pad tokens — data/__init__.py:326-355."""
    data = sorted(data, key=lambda d: len(d[1]), reverse=True)
    vids = torch.stack([d[0] for d in data], 0)
    lengths = [len(d[1]) for d in data]
    toks = torch.zeros(len(data), max(lengths), dtype=torch.long)
    for i, d in enumerate(data):
        toks[i, :lengths[i]] = d[1].long()
    return vids, toks, lengths


def my_dataset(data=None, vocab=None, anno=None, transform=None, random_frames=False, num_frames=16, use_lmdb=False,
               **synthetic_args):
    """Factory named by config/*.json (`"class": "txt2vid.data.my_dataset"`, data/__init__.py:376-377): a
    frame-folder `Dataset` when `data` is a directory (captions pickle = `--anno`); `data="synthetic"` (or a
    path that does not exist, e.g. the reference's own config/synth.json on a machine without the files)
    yields the synthetic clips, with `size= channels= length= seed=` forwarded."""
    import os
    if data is not None and data != 'synthetic' and os.path.isdir(str(data)):
        if anno is None:
            raise ValueError('a frame-folder dataset needs the captions pickle (--anno)')
        if transform is None:
            transform = default_transform([synthetic_args.get('size', 64)], synthetic_args.get('channels', 3))
        return Dataset(video_dir=data, vocab=vocab, captions=anno, transform=transform, random_frames=random_frames,
                       num_frames=num_frames, use_lmdb=use_lmdb)
    return SyntheticMovingDigits(num_frames=num_frames, vocab=vocab, **synthetic_args)


def get_loader(dset=None, batch_size=64, val=False, num_workers=0, has_captions=True, rank=0, world=1, seed=0, device=None):
    """data/__init__.py:379-383. With `world` > 1 every rank reads its own 1/world of each epoch's
    permutation (one process per GPU; the reference's single-process DataParallel split one big batch).
    Synthetic clips with `on_device` set (config: `"args": {"data": "synthetic", "on_device": true}`) and a CUDA `device` are
    generated in HBM by `DeviceSyntheticLoader` instead of on host workers."""
    if isinstance(dset, SyntheticMovingDigits) and getattr(dset, 'on_device', False) and device is not None and \
            torch.device(device).type == 'cuda':
        return DeviceSyntheticLoader(dset, batch_size, device, shuffle=not val)
    sampler = None
    if world > 1 and not isinstance(dset, SyntheticMovingDigits):     # synthetic clips already differ per rank (seed+rank)
        sampler = torch.utils.data.distributed.DistributedSampler(dset, num_replicas=world, rank=rank, shuffle=not val,
                                                                  seed=int(seed), drop_last=True)
    return torch.utils.data.DataLoader(dset, batch_size=batch_size, shuffle=(not val) and sampler is None, sampler=sampler,
                                       num_workers=num_workers, collate_fn=collate_fn, drop_last=True, pin_memory=True)


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
