"""Text auto-encoder pre-training entry point — the role of txt2vid/train/txt.py:89-235 (same command line, same iteration, same
checkpoint dict `{'optim', 'txt'}` that `train/gan.py --sent_weights` loads). Encoder and decoder run on the differentiable HIP
kernels (`Seq2Seq.differentiable(True)`: `t2v_lstm_train_step[_bwd]`, `t2v_embedding_bwd`, `t2v_xent_*`, `t2v_argmax_rows`, the
conv GEMMs for the input / vocabulary projections), Adam on the multi-tensor kernel. tensorboardX is optional (absent offline).

Layout: `CaptionSet` (token-id tensors of a sentence list), `pad_batch` (the loader's collate), `pretrain_loss` (one iteration's
loss), `Pretrainer` (optimiser step, validation, checkpoints, console log), `main`."""
import random
import sys

import torch

from ..models.txt.basic import Seq2Seq
from ..optim import Adam
from ..util.cli import parser_from
from ..util.dir import ensure_exists
from ..util.log import status
from ..util.metrics import RollingAvg
from ..util.pick import load
from .setup import setup

# txt.py:208-235 (flags, types, defaults); the last three rows are new: the reference hard-codes 50 / 500 and has no iteration cap
FLAGS = """
data str - !            # pickled {video: [sentence, ...]}
vocab str - !           # pickled Vocab
out str - !             # checkpoint directory
weights str -
test flag
separate_decoder flag
epoch int 5
batch_size int 64
lr float 0.001
beta1 float 0.9
beta2 float 0.999
seed int -
cuda flag
workers int 2
teacher_force float 0.5
max_seq_len int 10
log_period int 50
save_model_period int 500
max_iters int 0
"""


class CaptionSet(torch.utils.data.Dataset):
    """Sentences -> float tensors of token ids (txt.py:21-42). `sent_path`: pickle of {video: [sentences]}, flattened."""

    def __init__(self, vocab=None, sent_path=None, sents=None):
        if vocab is None:
            raise ValueError('a Vocab is required')
        self.vocab, self.sent_path = vocab, sent_path
        if sent_path is None:
            if sents is None:
                raise ValueError('give sent_path or sents')
            self.sents = list(sents)
        else:
            per_video = load(sent_path)
            self.sents = [s for key in per_video for s in per_video[key]]

    def __len__(self):
        return len(self.sents)

    def __getitem__(self, idx):
        ids = [self.vocab(tok) for tok in self.vocab.tokenize(self.sents[idx])]
        return torch.tensor(ids, dtype=torch.float32)


SentenceDataset = CaptionSet         # the reference's name for it


def pad_batch(items):
    """Collate (txt.py:45-53): longest first, zero padding -> (tokens [B, Lmax] int64, lengths)."""
    items = sorted(items, key=len, reverse=True)
    lengths = [int(t.numel()) for t in items]
    tokens = torch.zeros((len(items), lengths[0]), dtype=torch.int64)
    for row, (t, n) in enumerate(zip(items, lengths)):
        tokens[row, :n] = t.to(torch.int64)
    return tokens, lengths


collate_fn = pad_batch


def padded_targets(sent, lengths):
    """The targets of txt.py:166-167 (a pack -> pad round trip of the token matrix): tokens up to each length, 0 beyond, width
    lengths[0]. Index preparation on the host; returns a host int64 tensor."""
    width = int(lengths[0])
    keep = torch.arange(width).unsqueeze(0) < torch.tensor([int(n) for n in lengths]).unsqueeze(1)
    return sent.detach().cpu()[:, :width] * keep


def pretrain_loss(seq2seq, sent, lengths, teacher_force, reduction='mean'):
    """One iteration's loss (txt.py:160-172): encode, roll the decoder out from the encoder's state for lengths[0] positions,
    cross entropy over every (sample, position) — padding positions count as class 0. Returns (loss, decoded symbols)."""
    from .. import functional as TF
    state = seq2seq.encode(sent, lengths=lengths)[1]
    logits, symbols = seq2seq.decode(true_inputs=sent, initial_hidden=state, max_seq_len=lengths[0], teacher_force=teacher_force)
    rows = logits.shape[0] * logits.shape[1]
    return TF.cross_entropy(logits.view(rows, -1), padded_targets(sent, lengths).reshape(-1), reduction=reduction), symbols


def evaluate(batches, seq2seq, device, vocab, debug=False):
    """Validation / test loss (txt.py:55-87): greedy decoding, summed cross entropy divided by the number of sentences."""
    seq2seq.eval()
    total, count = 0.0, 0
    with torch.no_grad():
        for sent, lengths in batches:
            sent = sent.to(device)
            value, symbols = pretrain_loss(seq2seq, sent, lengths, False, reduction='sum')
            total += float(value)
            count += int(sent.shape[0])
            if debug:
                print('real words=', vocab.to_words(sent[-1]))
                print('predicted words=', vocab.to_words(symbols[-1]))
                print('loss=', float(value))
    seq2seq.train()
    return total / max(1, count)


class Pretrainer(object):
    """Optimiser step + the periodic work of txt.py:176-205 (validation and checkpoint every `save_model_period` iterations, a
    console line every `log_period`)."""

    def __init__(self, seq2seq, optimizer, vocab, device, args, val_batches):
        self.net, self.opt, self.vocab, self.device, self.args = seq2seq, optimizer, vocab, device, args
        self.val_batches = val_batches
        self.rolling = RollingAvg(window_size=args.log_period)
        self.iteration, self.val_loss = 0, -1
        try:
            from tensorboardX import SummaryWriter
            self.writer = SummaryWriter()
        except ImportError:
            self.writer = None

    def _scalar(self, tag, value):
        if self.writer is not None:
            self.writer.add_scalar(tag, value, self.iteration)

    def step(self, sent, lengths):
        sent = sent.to(self.device)
        self.net.zero_grad()
        forced = random.uniform(0, 1) <= self.args.teacher_force
        loss, symbols = pretrain_loss(self.net, sent, lengths, forced)
        loss.backward()
        self.opt.step()
        value = float(loss.detach())
        self.rolling.update(value)
        self._scalar('data/train_loss', value)
        self.iteration += 1
        return sent, symbols

    def checkpoint(self):
        self.val_loss = evaluate(self.val_batches, self.net, self.device, self.vocab)
        self._scalar('data/val_loss', self.val_loss)
        path = '%s/iter_%d_loss_%.4f_val_%.4f' % (self.args.out, self.iteration, self.rolling.get(), self.val_loss)
        print('saving to: %s' % path)
        # the whole module is pickled (reference layout, train/txt.py:185); it is saved in forward-only mode so that
        # `train/gan.py --sent_weights` gets an encoder on the capturable kernels, and switched back for the next iteration
        self.net.differentiable(False)
        try:
            torch.save({'optim': self.opt, 'txt': self.net}, path)
        finally:
            self.net.differentiable(True)

    def report(self, epoch, i, n_batches, sent, symbols):
        print('real words=', self.vocab.to_words(sent[0]))
        print('predicted words=', self.vocab.to_words(symbols[0]))
        print('[%d/%d][%d/%d] Loss: %.4f (val = %.4f)' % (epoch, self.args.epoch, i, n_batches, self.rolling.get(), self.val_loss))


def split_sentences(sents):
    """txt.py:117-127: shuffle, then an independent uniform draw per sentence: <= 0.8 train, <= 0.9 validation, else test."""
    random.shuffle(sents)
    parts = ([], [], [])
    for s in sents:
        r = random.uniform(0, 1)
        parts[0 if r <= 0.8 else 1 if r <= 0.9 else 2].append(s)
    if not all(parts):
        raise AssertionError('train / val / test split left a part empty (%d / %d / %d sentences)' % tuple(len(p) for p in parts))
    return parts


def main(args):
    seed, device = setup(args)
    ensure_exists(args.out)
    vocab = load(args.vocab)
    seq2seq = Seq2Seq(vocab_size=len(vocab), separate_decoder=args.separate_decoder).to(device)
    optimizer = Adam(seq2seq.parameters(), lr=args.lr, betas=(args.beta1, args.beta2))
    if args.weights:
        status('Loading model')
        from ..util.reflection import alias_reference_modules
        alias_reference_modules()
        saved = torch.load(args.weights, weights_only=False)
        seq2seq = saved['txt'].to(device) if 'txt' in saved else seq2seq
        optimizer = saved.get('optim', optimizer)
    seq2seq.differentiable(True)

    everything = CaptionSet(vocab=vocab, sent_path=args.data)
    train_s, val_s, test_s = split_sentences(everything.sents)
    for name, part in (('Train', train_s), ('Val', val_s), ('Test', test_s)):
        print('%s len = %d' % (name, len(part)))

    def loader(sents, shuffle):
        return torch.utils.data.DataLoader(CaptionSet(vocab=vocab, sents=sents), batch_size=args.batch_size, shuffle=shuffle,
                                           num_workers=args.workers, collate_fn=pad_batch)
    train_batches, val_batches, test_batches = loader(train_s, True), loader(val_s, False), loader(test_s, False)

    if args.test:
        status('Testing...')
        print('Test loss = %.4f' % evaluate(test_batches, seq2seq, device, vocab, debug=True))
        sys.exit(0)

    print('Teacher force prob = %.4f' % args.teacher_force)
    run = Pretrainer(seq2seq, optimizer, vocab, device, args, val_batches)
    for epoch in range(args.epoch):
        for i, (sent, lengths) in enumerate(train_batches):
            sent, symbols = run.step(sent, lengths)
            if run.iteration % args.save_model_period == 0:
                run.checkpoint()
            if run.iteration % args.log_period == 0:
                run.report(epoch, i, len(train_batches), sent, symbols)
            if args.max_iters and run.iteration >= args.max_iters:
                return run.rolling.get()
    return run.rolling.get()


def build_parser():
    return parser_from(FLAGS)


if __name__ == '__main__':
    main(build_parser().parse_args())
