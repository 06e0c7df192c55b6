"""CPU: host-side pieces of the text pre-training entry point (txt2vid_amd/train/txt.py) and the table-driven flag parser —
no kernel is called here (the product path has no CPU compute; tests/test_txt_gpu.py covers it on the MI355X)."""
import pickle
import random

import pytest
import torch

from txt2vid_amd.data import build_vocab
from txt2vid_amd.train import txt as TT
from txt2vid_amd.util.cli import parser_from


def test_flag_table_types_defaults_and_required():
    p = parser_from("""
    data str - !
    steps int 5
    lr float 0.001
    sizes ints 8 16
    names strs -
    fast flag            # trailing comment
    """)
    a = p.parse_args(['--data', 'x'])
    assert (a.data, a.steps, a.lr, a.sizes, a.names, a.fast) == ('x', 5, 0.001, [8, 16], None, False)
    a = p.parse_args(['--data', 'x', '--sizes', '4', '--names', 'a', 'b', '--fast', '--lr', '2e-4'])
    assert a.sizes == [4] and a.names == ['a', 'b'] and a.fast is True and a.lr == 2e-4
    with pytest.raises(SystemExit):
        p.parse_args([])                     # --data is required


def test_pretraining_parser_matches_reference_surface():
    """Flag names, types and defaults of txt2vid/train/txt.py:208-235."""
    a = TT.build_parser().parse_args(['--data', 'd', '--vocab', 'v', '--out', 'o'])
    want = dict(weights=None, test=False, separate_decoder=False, epoch=5, batch_size=64, lr=0.001, beta1=0.9, beta2=0.999,
                seed=None, cuda=False, workers=2, teacher_force=0.5, max_seq_len=10)
    for k, v in want.items():
        assert getattr(a, k) == v, k


def test_caption_set_and_padding(tmp_path):
    sents = {'a': ['red digit moves left.', 'two'], 'b': ['blue digit moves up and down.']}
    vocab = build_vocab([s for v in sents.values() for s in v])
    with open(tmp_path / 's.pkl', 'wb') as f:
        pickle.dump(sents, f)
    ds = TT.CaptionSet(vocab=vocab, sent_path=str(tmp_path / 's.pkl'))
    assert len(ds) == 3 and TT.SentenceDataset is TT.CaptionSet
    items = [ds[i] for i in range(3)]
    assert all(t.dtype == torch.float32 for t in items)
    assert int(items[0][0]) == vocab(vocab.START) and int(items[0][-1]) == vocab(vocab.END)
    tokens, lengths = TT.pad_batch(list(items))
    assert lengths == sorted(lengths, reverse=True) and tokens.dtype == torch.int64 and tokens.shape == (3, lengths[0])
    for row, n in zip(tokens, lengths):
        assert (row[n:] == 0).all() and (row[:n] != 0).all()
    # the pack -> pad round trip of the reference zero-fills whatever sits beyond each length
    dirty = tokens.clone()
    dirty[1, lengths[1]:] = 7
    tg = TT.padded_targets(dirty, lengths)
    assert tg.shape == (3, lengths[0]) and torch.equal(tg, tokens)
    with pytest.raises(ValueError):
        TT.CaptionSet(vocab=None, sents=['x'])


def test_split_sentences_is_seeded_and_complete():
    sents = ['s%d' % i for i in range(200)]
    random.seed(5)
    a = TT.split_sentences(list(sents))
    random.seed(5)
    b = TT.split_sentences(list(sents))
    assert a == b
    assert sorted(a[0] + a[1] + a[2]) == sorted(sents)
    assert len(a[0]) > len(a[1]) and len(a[0]) > len(a[2])
    random.seed(1)
    with pytest.raises(AssertionError):
        TT.split_sentences(['only'])
