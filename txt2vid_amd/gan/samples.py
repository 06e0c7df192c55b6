"""Sample dumps — `save_frames` / `save_sentences` of txt2vid/gan/trainer.py:92-108 without torchvision:
a [b,C,T,H,W] batch becomes one PNG grid (one row per clip, `nrow = T`, 2-pixel padding, min/max
normalisation over the whole tensor like `torchvision.utils.save_image(normalize=True)`). `.png` paths go through the
built-in writer, any other extension through PIL (the reference's sampling path writes `.jpg`). Host-side I/O."""
import struct
import zlib

import numpy as np


def _png(path, img):
    """img: uint8 [H,W,3]."""
    h, w, _ = img.shape
    raw = b''.join(b'\x00' + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack('>I', len(data)) + tag + data
        return c + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def make_grid(frames, nrow, padding=2):
    """frames: float [n,C,H,W] in [0,1] -> float [3, gh, gw] (torchvision.utils.make_grid layout)."""
    n, c, h, w = frames.shape
    if c == 1:
        frames = np.repeat(frames, 3, axis=1)
    xmaps = min(nrow, n)
    ymaps = int(np.ceil(n / float(xmaps)))
    H, W = h + padding, w + padding
    grid = np.zeros((3, H * ymaps + padding, W * xmaps + padding), dtype=np.float32)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * H + padding:y * H + padding + h, x * W + padding:x * W + padding + w] = frames[k]
            k += 1
    return grid


def save_frames(x, path=None, channel_first=True, is_images=False):
    """x: [b,C,T,H,W] device or host tensor (channel_first) -> PNG grid, one clip per row."""
    a = x.detach().float().cpu().numpy()
    if is_images:
        frames, nrow = a, 8
    else:
        if channel_first:
            a = a.transpose(0, 2, 1, 3, 4)                      # [b,T,C,H,W]
        nrow = a.shape[1]
        frames = a.reshape((-1,) + a.shape[2:])
    lo, hi = float(frames.min()), float(frames.max())
    frames = (frames - lo) / max(hi - lo, 1e-5)
    grid = make_grid(frames, nrow)
    img = (np.clip(grid, 0, 1) * 255 + 0.5).astype(np.uint8).transpose(1, 2, 0)
    if str(path).lower().endswith('.png'):
        _png(path, img)
    else:                                   # the reference's `<h>x<w>_<i>_<j>.jpg`: the format follows the extension
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(img)).save(path)


def save_sentences(captions, path=None, vocab=None):
    with open(path, 'w') as f:
        for cap in captions:
            f.write(vocab.to_words(cap.tolist() if hasattr(cap, 'tolist') else cap))
            f.write('\n')
