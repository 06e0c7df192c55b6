"""CPU: host-side logic of the drop-in surface (no kernels): reflection, init parity with the reference,
state_dict layout, random-draw bookkeeping, data contract, checkpoint key matching."""
import random

import numpy as np
import pytest
import torch

from oracle import tganv2_oracle as O


def test_reflection_aliases_reference_names():
    from txt2vid_amd.util.reflection import create_object, get_class
    from txt2vid_amd.models.tganv2_cond.gen import MultiScaleGen
    assert get_class('txt2vid.models.tganv2_cond.gen.MultiScaleGen') is MultiScaleGen
    loss = create_object('txt2vid.gan.losses.RSGANLoss')
    assert loss.__class__.__name__ == 'RSGANLoss'
    d = create_object({'class': 'txt2vid.models.tganv2.discrim.MultiScaleDiscrim', 'args': {'num_channels': 1}}, cond_dim=0)
    assert len(d.sub_discrims) == 4
    ds = create_object({'class': 'txt2vid.data.my_dataset', 'args': {'data': '/nonexistent/videos', 'num_frames': 16}}, vocab=None)
    v, c = ds[0]
    assert tuple(v.shape) == (16, 1, 64, 64) and len(c) == 8 and float(v.min()) == -1.0


@pytest.mark.parametrize('which', ['uncond', 'cond'])
def test_state_dict_layout_matches_reference(which):
    """Checkpoint interchange contract: key names and shapes equal the reference's."""
    if which == 'uncond':
        from txt2vid_amd.models.tganv2.gen import MultiScaleGen
        from txt2vid_amd.models.tganv2.discrim import MultiScaleDiscrim
        g, d = MultiScaleGen(width=64, height=64, num_channels=1), MultiScaleDiscrim(num_channels=1)
        gs, ds_ = O.gen_shapes(num_channels=1), O.resnet3d_shapes('single_discrim.', 1, 64, 0)
    else:
        from txt2vid_amd.models.tganv2_cond.gen import MultiScaleGen
        from txt2vid_amd.models.tganv2_cond.discrim import MultiScaleDiscrim
        g, d = MultiScaleGen(width=64, height=64, num_channels=1, cond_dim=256), MultiScaleDiscrim(num_channels=1, cond_dim=256)
        gs = O.gen_shapes(num_channels=1, cond_dim=256, cond_variant=True)
        ds_ = O.resnet3d_shapes('single_discrim.module.', 1, 64, 256)
    for mod, shapes in ((g, gs), (d, ds_)):
        sd = mod.state_dict()
        assert set(sd.keys()) == set(shapes.keys())
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), k


def test_xavier_init_reproduces_reference(golden):
    """seed 100 -> construct G then D -> init(..., 'xavier'): per-key checksums equal those recorded from
    the reference (train/setup.py:7-14, train/gan.py:60-70, util/torch/init.py:4-39)."""
    from txt2vid_amd.models.tganv2.gen import MultiScaleGen
    from txt2vid_amd.models.tganv2.discrim import MultiScaleDiscrim
    from txt2vid_amd.util.torch.init import init
    g = golden('init_xavier')
    random.seed(100)
    np.random.seed(100)
    torch.manual_seed(100)
    gen = MultiScaleGen(width=64, height=64, num_channels=1)
    dis = MultiScaleDiscrim(num_channels=1)
    init(gen, 'xavier')
    init(dis, 'xavier')
    for tag, m in (('G', gen), ('D', dis)):
        sd = m.state_dict()
        for k, s, a in zip([str(k) for k in g[tag + '_keys']], g[tag + '_sum'], g[tag + '_abs']):
            assert abs(float(sd[k].double().sum()) - s) <= 1e-6 * max(1.0, a), k
            assert abs(float(sd[k].double().abs().sum()) - a) <= 1e-6 * max(1.0, a), k


def test_gen_perm_and_metrics():
    from txt2vid_amd.util.misc import gen_perm
    from txt2vid_amd.util.metrics import RollingAvg
    np.random.seed(0)
    for n in (2, 3, 8):
        p = gen_perm(n)
        assert sorted(p.tolist()) == list(range(n)) and not (p == np.arange(n)).all()
    with pytest.raises(ValueError):
        gen_perm(1)
    np.random.seed(3)
    a = gen_perm(6)
    np.random.seed(3)
    b = O.gen_perm(6)
    assert (a == b).all()
    r = RollingAvg(window_size=3)
    for v in (1, 2, 3, 4):
        r.update(v)
    assert r.get() == 3.0


def test_draw_sources_follow_the_reference_order():
    """HostDraws (eager) and StaticDraws (graph replay) consume the CPU generator identically, and the
    cumulative phases equal composing `x[::2, :, bt::2]` level by level (trainer.py:157-158)."""
    from txt2vid_amd import functional as TF
    torch.manual_seed(9)
    h = TF.HostDraws()
    t0s = [t for t, _ in h.multiscale_t0(4)]
    z = torch.randn(4, 8)
    ph = [int(torch.randint(2, (1,))) for _ in range(3)]
    al = [torch.rand(b, 1, 1, 1, 1).reshape(b) for b in (4, 2, 1, 1)]
    # same seed, oracle-style composition of Subsample
    torch.manual_seed(9)
    x = torch.arange(16).view(1, 1, 16, 1, 1).float()
    lv = []
    for i in range(4):
        lv.append(x)
        x, _ = O.subsample(x)
    for i in range(4):
        assert int(lv[i][0, 0, 0, 0, 0]) == t0s[i]
        assert lv[i].shape[2] == 16 // 2 ** i
    # StaticDraws needs pinned memory + a device; exercised on the GPU in test_models_gpu.py


def test_data_contract():
    from txt2vid_amd.data import SyntheticMovingDigits, collate_fn, Vocab
    ds = SyntheticMovingDigits(length=8)
    vids, toks, lens = collate_fn([ds[i] for i in range(4)])
    assert tuple(vids.shape) == (4, 16, 1, 64, 64) and toks.dtype == torch.long and lens == sorted(lens, reverse=True)
    assert float(vids.max()) <= 1.0 and float(vids.min()) == -1.0
    v = Vocab()
    assert len(v) == 21 and v.to_words(toks[0]).startswith('<start> digit')
    a, _ = ds[2]
    b, _ = SyntheticMovingDigits(length=8)[2]
    assert torch.equal(a, b)


def test_checkpoint_key_styles_interchange():
    """`single_discrim.*` (uncond) <-> `single_discrim.module.*` (cond wrapper) — SURVEY §5."""
    from txt2vid_amd.gan.cond_gan import _match_keys
    want = ['single_discrim.module.fc.weight', 'single_discrim.module.fc.bias']
    sd = {'single_discrim.fc.weight': 1, 'single_discrim.fc.bias': 2}
    out = _match_keys(sd, want)
    assert out == {'single_discrim.module.fc.weight': 1, 'single_discrim.module.fc.bias': 2}
    assert _match_keys({k: 0 for k in want}, want) == {k: 0 for k in want}


def test_cli_flags_match_reference():
    import argparse
    from txt2vid_amd.gan.trainer import add_params_to_parser
    p = add_params_to_parser(argparse.ArgumentParser())
    a = p.parse_args(['--gp_lambda', '.5', '--subsample_input', '--no_mean_discrim_loss'])
    assert a.gp_lambda == 0.5 and a.subsample_input and a.no_mean_discrim_loss is False and a.no_mean_gen_loss is True


def test_reference_pickled_sentence_encoder_resolves(tmp_path):
    """`--sent_weights`: the reference saves `{'optim': ..., 'txt': seq2seq}` with the objects pickled whole under
    `txt2vid.models.txt.basic` (train/txt.py:185); with the module aliases the same file loads into the drop-in class."""
    import pickle
    import sys
    import torch
    from txt2vid_amd.models.txt.basic import Seq2Seq
    from txt2vid_amd.util.reflection import alias_reference_modules
    m = Seq2Seq(vocab_size=11)
    # (a text protocol: binary protocols prefix the module path with its length, which the rename would break)
    blob = pickle.dumps({'txt': m, 'optim': None}, protocol=0).replace(b'txt2vid_amd.models.txt.basic', b'txt2vid.models.txt.basic')
    for k in [k for k in sys.modules if k == 'txt2vid' or k.startswith('txt2vid.')]:
        del sys.modules[k]
    alias_reference_modules()
    got = pickle.loads(blob)['txt']
    assert isinstance(got, Seq2Seq) and got.encoder.lstm.weight_hh_l0.shape == m.encoder.lstm.weight_hh_l0.shape
    assert torch.equal(got.encoder.embed.weight, m.encoder.embed.weight)


def _purge_reference_modules():
    import sys
    for k in [k for k in sys.modules if k == 'txt2vid' or k.startswith('txt2vid.')]:
        del sys.modules[k]


def test_sentence_encoder_pickled_by_the_real_reference(golden):
    """tests/golden/ref_seq2seq.pt was written by the REFERENCE's own classes (`torch.save({'optim': Adam, 'txt': Seq2Seq})`,
    train/txt.py:185; make_golden.py `seq2seq_pickle`; parameter values compacted). Unpickling never runs this build's
    __init__, so the instance lacks every attribute only this build sets: it must still resolve to the drop-in class and be
    in forward-only mode (`with_grad` was an instance attribute)."""
    import os
    from conftest import GOLDEN
    from txt2vid_amd.models.txt.basic import Seq2Seq, RecurrentModel
    from txt2vid_amd.util.reflection import alias_reference_modules
    _purge_reference_modules()
    alias_reference_modules()
    ck = torch.load(os.path.join(GOLDEN, 'ref_seq2seq.pt'), weights_only=False, map_location='cpu')
    assert set(ck) == {'optim', 'txt'}
    txt = ck['txt']
    assert isinstance(txt, Seq2Seq) and isinstance(txt.encoder, RecurrentModel) and txt.decoder is txt.encoder
    assert 'with_grad' not in txt.encoder.__dict__ and txt.encoder.with_grad is False
    assert txt.encoder.encoding_size == 256 and txt.encoder.hidden_size == 128 and txt.encoder.bi and txt.encoder.num_layers == 4
    want = Seq2Seq(vocab_size=21).state_dict()
    got = txt.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert got[k].shape == want[k].shape and got[k].dtype == want[k].dtype, k
    txt.differentiable(False)                                    # what train/gan.py does after loading
    assert txt.encoder.__dict__['with_grad'] is False
    assert isinstance(ck['optim'], torch.optim.Adam)


@pytest.mark.parametrize('which', ['uncond', 'cond'])
def test_checkpoint_written_by_the_real_reference_loads(which):
    """tests/golden/ref_checkpoint_{uncond,cond}.pt: the dict `trainer.train()` saves (trainer.py:269-279), built by the
    reference's CondGan.save_dict + torch.optim.Adam.state_dict after one optimiser step (make_golden.py `checkpoint`; one
    element per tensor kept). `CondGan.load_from_dict` + both optimisers' `load_state_dict` must take it as is — the
    reference's own resume sequence, train/gan.py:113-127 — and every tensor must arrive (closes SURVEY §8 f3)."""
    import os
    from conftest import GOLDEN
    from txt2vid_amd.gan.cond_gan import CondGan
    from txt2vid_amd.optim import Adam
    if which == 'uncond':
        from txt2vid_amd.models.tganv2.gen import MultiScaleGen
        from txt2vid_amd.models.tganv2.discrim import MultiScaleDiscrim
        gen, dis, txt = MultiScaleGen(width=64, height=64, num_channels=1), MultiScaleDiscrim(num_channels=1), None
    else:
        from txt2vid_amd.models.tganv2_cond.gen import MultiScaleGen
        from txt2vid_amd.models.tganv2_cond.discrim import MultiScaleDiscrim
        from txt2vid_amd.models.txt.basic import Seq2Seq
        gen = MultiScaleGen(width=64, height=64, num_channels=1, cond_dim=256)
        dis = MultiScaleDiscrim(num_channels=1, cond_dim=256)
        txt = Seq2Seq(vocab_size=21)
    ck = torch.load(os.path.join(GOLDEN, 'ref_checkpoint_%s.pt' % which), weights_only=False, map_location='cpu')
    assert set(ck) == {'optG', 'optD', 'gen', 'video'} | ({'cond'} if txt is not None else set())
    style = 'single_discrim.module.' if which == 'cond' else 'single_discrim.'
    assert all(k.startswith(style) for k in ck['video'])
    gan = CondGan(gen=gen, discrims=[dis], cond_encoder=txt, discrim_names=['video'])
    for m in (gen, dis):
        for p in m.parameters():
            p.data.fill_(123.0)
    gan.load_from_dict(ck)
    for name, mod in (('gen', gen), ('video', dis)) + ((('cond', txt),) if txt is not None else ()):
        sd = mod.state_dict()
        assert set(sd.keys()) == set(ck[name].keys()), (name, set(sd) ^ set(ck[name]))
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(ck[name][k].shape), k
            if v.dtype.is_floating_point:
                assert torch.equal(v, ck[name][k].contiguous()), (name, k)        # every tensor arrived
    optD = Adam([{'params': dis.parameters()}], lr=1e-4, betas=(0.5, 0.9))
    optG = Adam([{'params': gen.parameters()}], lr=1e-4, betas=(0.5, 0.9))
    optD.load_state_dict(ck['optD'])
    optG.load_state_dict(ck['optG'])
    for opt, mod, key in ((optD, dis, 'optD'), (optG, gen, 'optG')):
        grp = opt.param_groups[0]
        assert grp['lr'] == 2e-4 and tuple(grp['betas']) == (0.5, 0.999)       # the checkpoint's hyper-parameters win, as in torch
        params = list(mod.parameters())
        assert len(opt.state) == len(params) == len(ck[key]['state'])
        for i, p in enumerate(params):
            st, ref = opt.state[p], ck[key]['state'][i]
            assert int(st['step']) == 1 and st['exp_avg'].shape == p.shape and st['exp_avg_sq'].shape == p.shape
            assert torch.equal(st['exp_avg'], ref['exp_avg'].contiguous()), (key, i)
            for k in ('exp_avg', 'exp_avg_sq'):        # dense memory of their own IN THE PARAMETER'S LAYOUT: the kernels walk p, g, m, v with one index
                assert st[k].stride() == p.stride() and st[k].untyped_storage().nbytes() >= st[k].numel() * 4, (key, i, k)


def test_tap_major_master_weights_keep_values_draw_order_and_state_dict():
    """The ConvLSTM's eight [1024,1024,3,3] weights are stored tap-major in memory (models/conv_lstm.py) while shape, values,
    `init()` draw order and state_dict stay those of the reference's dense weights."""
    import torch.nn as nn
    from txt2vid_amd import functional as TF
    from txt2vid_amd.models.conv_lstm import ConvLSTMCell
    from txt2vid_amd.util.torch.init import init
    torch.manual_seed(7)
    cell = ConvLSTMCell(8, 8, 3)
    torch.manual_seed(7)
    plain = [nn.Conv2d(8, 8, 3, 1, 1, bias=b) for b in (True, False) * 4]          # same constructor order: Wxi Whi Wxf Whf ...
    names = ('Wxi', 'Whi', 'Wxf', 'Whf', 'Wxc', 'Whc', 'Wxo', 'Who')
    for n, ref in zip(names, plain):
        w = getattr(cell, n).weight
        assert TF.is_tap_major(w) and tuple(w.shape) == (8, 8, 3, 3) and torch.equal(w, ref.weight)
    torch.manual_seed(11)
    init(cell, 'xavier')
    torch.manual_seed(11)
    for ref in plain:                                   # children-first, left to right = the order above
        nn.init.xavier_normal_(ref.weight)
    for n, ref in zip(names, plain):
        assert torch.equal(getattr(cell, n).weight, ref.weight), n
    sd = cell.state_dict()
    assert list(sd)[:3] == ['Wxi.weight', 'Wxi.bias', 'Whi.weight']
    dense = {k: v.contiguous() for k, v in sd.items()}  # what a reference checkpoint holds
    cell2 = ConvLSTMCell(8, 8, 3)
    cell2.load_state_dict(dense)
    assert all(torch.equal(a, b) for a, b in zip(cell.state_dict().values(), cell2.state_dict().values()))
    assert TF.is_tap_major(cell2.Wxi.weight)
    rows = TF.tap_rows(cell.Wxi.weight)
    assert rows.shape == (9, 64) and torch.equal(rows[4].view(8, 8), cell.Wxi.weight[:, :, 1, 1])


def test_sample_grid_writer_pixel_level(tmp_path):
    """`samples.save_frames` = torchvision.utils.save_image(frames, normalize=True, nrow=T) of trainer.py:92-101: min/max
    normalisation over the whole tensor, one clip per row, 2-pixel zero padding, round-half-up to uint8 — decoded back
    from the PNG and compared pixel by pixel with the layout computed independently here; grey clips become 3 equal channels."""
    from PIL import Image
    from txt2vid_amd.gan.samples import save_frames
    g = torch.Generator()
    g.manual_seed(5)
    for C_ in (1, 3):
        x = torch.randn(3, C_, 4, 6, 5, generator=g) * 2.0 - 0.3                # [b,C,T,H,W]
        path = tmp_path / ('grid%d.png' % C_)
        save_frames(x, str(path))
        img = np.asarray(Image.open(path).convert('RGB')).astype(np.int64)       # [gh, gw, 3]
        b, _, T, H, W = x.shape
        assert img.shape == (b * (H + 2) + 2, T * (W + 2) + 2, 3)
        lo, hi = float(x.min()), float(x.max())
        want = np.zeros(img.shape, dtype=np.int64)
        for n in range(b):
            for t in range(T):
                tile = ((x[n, :, t] - lo) / (hi - lo)).clamp(0, 1).mul(255).add(0.5).floor().numpy().astype(np.int64)
                tile = np.repeat(tile, 3, axis=0) if C_ == 1 else tile
                want[n * (H + 2) + 2:n * (H + 2) + 2 + H, t * (W + 2) + 2:t * (W + 2) + 2 + W, :] = tile.transpose(1, 2, 0)
        assert np.array_equal(img, want)
    save_frames(x, str(tmp_path / 'grid.jpg'))                                   # the sampling path's extension: PIL writes JPEG
    assert Image.open(tmp_path / 'grid.jpg').format == 'JPEG'


def test_adam_refuses_state_with_foreign_strides():
    """A moment assigned straight into `opt.state` with stride-0 (or any non-parameter) strides would be walked as a dense array
    by the kernel (the round-2 GPU fault): `step()` raises before launching anything; `load_state_dict` re-materialises instead."""
    from txt2vid_amd.optim import Adam, SGD
    p = torch.nn.Parameter(torch.zeros(4, 3))
    p.grad = torch.ones(4, 3)
    opt = Adam([p], lr=1e-3)
    opt.state[p] = {'step': 1, 'exp_avg': torch.zeros(1, 1).expand(4, 3), 'exp_avg_sq': torch.zeros(4, 3)}
    with pytest.raises(ValueError, match='strides'):
        opt.step()
    assert opt.state[p]['step'] == 1                       # nothing advanced
    opt.state[p]['exp_avg'] = torch.zeros(4, 3, dtype=torch.float64)
    with pytest.raises(ValueError, match='does not match'):
        opt.step()
    opt.state[p]['exp_avg'] = torch.zeros(8, 3)[::2]       # right shape, a strided slice of somebody else's buffer
    with pytest.raises(ValueError, match='strides'):
        opt.step()
    sgd = SGD([p], lr=1e-3, momentum=0.5)
    sgd.state[p] = {'momentum_buffer': torch.zeros(1, 1).expand(4, 3)}
    with pytest.raises(ValueError, match='strides'):
        sgd.step()
    # the supported way in: load_state_dict densifies
    opt2 = Adam([p], lr=1e-3)
    sd = {'state': {0: {'step': 1, 'exp_avg': torch.zeros(1, 1).expand(4, 3), 'exp_avg_sq': torch.zeros(1, 1).expand(4, 3)}},
          'param_groups': opt2.state_dict()['param_groups']}
    opt2.load_state_dict(sd)
    assert opt2.state[p]['exp_avg'].stride() == p.stride() and opt2.state[p]['exp_avg'].is_contiguous()


def test_rng_state_round_trip_continues_the_draw_sequence():
    """Checkpoints carry the three host generators' states (`rng_state`): restoring them continues the sequence of z / phase /
    alpha (torch), caption permutation (numpy) and `random` draws exactly where the saving run stood."""
    import io
    from txt2vid_amd.train.setup import get_rng_state, set_rng_state, set_seed
    set_seed(77)
    torch.randn(5), np.random.permutation(7), random.random()
    blob = io.BytesIO()
    torch.save({'rng_state': get_rng_state(), 'iteration': 3}, blob)          # through the pickle the checkpoint uses
    want = (torch.randn(4), torch.randint(2, (3,)), np.random.permutation(9), random.random())
    set_seed(5)
    blob.seek(0)
    set_rng_state(torch.load(blob, weights_only=False)['rng_state'])
    got = (torch.randn(4), torch.randint(2, (3,)), np.random.permutation(9), random.random())
    assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1]) and (want[2] == got[2]).all() and want[3] == got[3]
    set_rng_state(None)                                                        # reference-written files: no key, nothing happens
    set_rng_state({})


def test_grad_sink_overflow_keeps_the_first_bias_writer_a_store():
    """when a destination overflows its source list (or meets another bias) it is flushed; the incoming producer must
    keep `accumulate = False` for a bias nobody wrote yet this step, and the frozen sink refuses to grow / recycle."""
    from txt2vid_amd import functional as TF
    from txt2vid_amd._lib import WgradSrc, WGRAD_MAX_SRC
    sink = TF.GradSink([])
    flushed = []

    def fake_flush(dests):
        for d in dests:
            flushed.append((d.wid, d.bid, d.accum, d.accum_bias, len(d.srcs)))
            sink.pending.pop(d.wid, None)
            if d.bid is not None:
                sink.pending_bias.pop(d.bid, None)
    sink._flush = fake_flush
    w, b = torch.zeros(8), torch.zeros(2)
    for _ in range(WGRAD_MAX_SRC):                          # bias-less producers fill the source list
        sink.add_partial(w, w, False, None, None, False, WgradSrc(), None, 1, 2, 4)
    sink.add_partial(w, w, True, b, b, False, WgradSrc(), None, 1, 2, 4)     # first producer WITH the bias: overflow -> flush
    assert flushed == [(id(w), None, 0, 0, WGRAD_MAX_SRC)]
    d = sink.pending[id(w)]
    assert d.accum == 1 and d.accum_bias == 0              # weight slot was written by the flush; the bias slot was not
    # same bias on both sides of an overflow: now the flushed table did write it
    sink.pending.clear(), sink.pending_bias.clear(), flushed.clear()
    for _ in range(WGRAD_MAX_SRC):
        sink.add_partial(w, w, False, b, b, False, WgradSrc(), None, 1, 2, 4)
    sink.add_partial(w, w, True, b, b, False, WgradSrc(), None, 1, 2, 4)
    assert sink.pending[id(w)].accum_bias == 1
    # frozen: a captured graph reads the workspace
    sink.pending.clear(), sink.pending_bias.clear()
    sink.frozen, sink.ws_used = 1, 128
    with pytest.raises(RuntimeError, match='captured HIP graph'):
        sink.reset()


def test_arena_gather_is_stride_aware_for_tap_major_slots():
    """a dense gradient for a tap-major master weight must land in [kh][kw][Cout][Cin] memory order inside the arena
    (CPU path; the GPU copy path has its own test in test_dp_gpu.py)."""
    from txt2vid_amd import functional as TF
    from txt2vid_amd.dist import GradArena
    w = torch.nn.Parameter(TF.tap_major(torch.zeros(4, 3, 3, 3)))
    b = torch.nn.Parameter(torch.zeros(5))
    arena = GradArena([b, w], live_taps={w: [4]})
    g = torch.arange(4 * 3 * 9, dtype=torch.float32).view(4, 3, 3, 3)
    w.grad, b.grad = g.clone(), torch.ones(5)              # dense gradient (it bypassed the sink)
    arena.gather()
    v = arena.views()[1]
    assert torch.equal(v, g) and v.stride() == w.stride()
    off = arena.offsets[1]
    assert torch.equal(arena.flat[off + 4 * 12:off + 5 * 12].view(4, 3), g[:, :, 1, 1])          # tap 4 is one contiguous row
