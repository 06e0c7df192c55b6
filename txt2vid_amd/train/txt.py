"""Text auto-encoder pre-training entry point — the role of txt2vid/train/txt.py:89-235 (same flags, same loop, same checkpoint
dict `{'optim', 'txt'}` that `train/gan.py --sent_weights` loads). Encoder and decoder run on the differentiable HIP kernels
(`Seq2Seq.differentiable(True)`: `t2v_lstm_train_step[_bwd]`, `t2v_embedding_bwd`, `t2v_xent_*`, `t2v_argmax_rows`, the conv GEMMs
for the input / vocabulary projections), Adam on the multi-tensor kernel. tensorboardX is optional here (absent offline)."""
import argparse
import random
import sys

import torch

from ..models.txt.basic import Seq2Seq
from ..optim import Adam
from ..util.dir import ensure_exists
from ..util.log import status
from ..util.metrics import RollingAvg
from ..util.pick import load
from .setup import setup


class SentenceDataset(torch.utils.data.Dataset):
    """txt.py:21-42: `sent_path` = pickled {video: [sentences]}; items are FloatTensors of token ids."""

    def __init__(self, vocab=None, sent_path=None, sents=None):
        assert vocab is not None
        self.vocab = vocab
        self.sent_path = sent_path
        if sent_path is not None:
            temp = load(sent_path)
            self.sents = [s for x in temp for s in temp[x]]
        else:
            assert sents is not None
            self.sents = sents

    def __len__(self):
        return len(self.sents)

    def __getitem__(self, idx):
        return torch.Tensor([self.vocab(token) for token in self.vocab.tokenize(self.sents[idx])])


def collate_fn(data):
    """txt.py:45-53: sort by length (descending), zero-pad -> (tokens [B,Lmax] long, lengths)."""
    data.sort(key=lambda x: len(x), reverse=True)
    lengths = [len(sent) for sent in data]
    targets = torch.zeros(len(data), max(lengths)).long()
    for i, sent in enumerate(data):
        targets[i, :lengths[i]] = sent[:lengths[i]]
    return targets, lengths


def padded_targets(sent, lengths):
    """txt.py:166-167 (`pack_padded_sequence` -> `pad_packed_sequence` of the token matrix): tokens up to each length, 0 beyond,
    width lengths[0]. Host-side index preparation; returns a host int64 tensor."""
    L = int(lengths[0])
    t = sent.detach().cpu()[:, :L].clone()
    for b, n in enumerate(lengths):
        t[b, int(n):] = 0
    return t


def pretrain_loss(seq2seq, sent, lengths, teacher_force, reduction='mean'):
    """Loop body of txt.py:160-172: encode, decode from the encoder's state, cross entropy over every (sample, position)."""
    from .. import functional as TF
    _, hidden_states, _ = seq2seq.encode(sent, lengths=lengths)
    targets = padded_targets(sent, lengths)
    decoded, d_symbols = seq2seq.decode(true_inputs=sent, initial_hidden=hidden_states, max_seq_len=lengths[0],
                                        teacher_force=teacher_force)
    B, L, V = decoded.shape
    return TF.cross_entropy(decoded.view(B * L, V), targets.reshape(-1), reduction=reduction), d_symbols


def evaluate(split, seq2seq, device, vocab, debug=False):
    """txt.py:55-87: greedy decoding, summed cross entropy per example."""
    seq2seq.eval()
    loss, num_examples = 0.0, 0
    with torch.no_grad():
        for sent, lengths in split:
            sent = sent.to(device)
            temp, d_symbols = pretrain_loss(seq2seq, sent, lengths, False, reduction='sum')
            if debug:
                print('real words=', vocab.to_words(sent[-1]))
                print('predicted words=', vocab.to_words(d_symbols[-1]))
                print('loss=', float(temp))
            loss += float(temp)
            num_examples += sent.size(0)
    seq2seq.train()
    return loss / max(1, num_examples)


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
        temp = torch.load(args.weights, weights_only=False)
        if 'txt' in temp:
            seq2seq = temp['txt'].to(device)
        if 'optim' in temp:
            optimizer = temp['optim']
    seq2seq.differentiable(True)

    train, val, test = [], [], []
    data = SentenceDataset(vocab=vocab, sent_path=args.data)
    random.shuffle(data.sents)
    for i in range(len(data.sents)):
        r = random.uniform(0, 1)
        (train if r <= 0.8 else val if r <= 0.9 else test).append(data.sents[i])
    assert len(val) != 0 and len(test) != 0 and len(train) != 0
    data.sents = train
    train = data
    print('Train len = %d' % len(train))
    print('Val len = %d' % len(val))
    print('Test len = %d' % len(test))
    test = SentenceDataset(vocab=vocab, sents=test)
    val = SentenceDataset(vocab=vocab, sents=val)
    mk = torch.utils.data.DataLoader
    train_dataset = mk(train, batch_size=args.batch_size, shuffle=True, num_workers=args.workers, collate_fn=collate_fn)
    test_dataset = mk(test, batch_size=args.batch_size, shuffle=False, num_workers=args.workers, collate_fn=collate_fn)
    val_dataset = mk(val, batch_size=args.batch_size, shuffle=False, num_workers=args.workers, collate_fn=collate_fn)

    if args.test:
        status('Testing...')
        print('Test loss = %.4f' % evaluate(test_dataset, seq2seq, device, vocab, debug=True))
        sys.exit(0)

    log_window_period = args.log_period
    save_model_period = args.save_model_period
    rolling_loss = RollingAvg(window_size=log_window_period)
    try:
        from tensorboardX import SummaryWriter
        writer = SummaryWriter()
    except ImportError:
        writer = None
    print('Teacher force prob = %.4f' % args.teacher_force)

    iteration, val_loss = 0, -1
    for epoch in range(args.epoch):
        for i, (sent, lengths) in enumerate(train_dataset):
            sent = sent.to(device)
            seq2seq.zero_grad()
            teacher_force = random.uniform(0, 1) <= args.teacher_force
            loss, d_symbols = pretrain_loss(seq2seq, sent, lengths, teacher_force)
            loss.backward()
            optimizer.step()
            loss_v = float(loss)
            rolling_loss.update(loss_v)
            if writer is not None:
                writer.add_scalar('data/train_loss', loss_v, iteration)
            iteration += 1
            if iteration % save_model_period == 0:
                val_loss = evaluate(val_dataset, seq2seq, device, vocab)
                if writer is not None:
                    writer.add_scalar('data/val_loss', val_loss, iteration)
                where_to_save = '%s/iter_%d_loss_%.4f_val_%.4f' % (args.out, iteration, rolling_loss.get(), val_loss)
                print('saving to: %s' % where_to_save)
                torch.save({'optim': optimizer, 'txt': seq2seq}, where_to_save)
            if iteration % log_window_period == 0:
                print('real words=', vocab.to_words(sent[0]))
                print('predicted words=', vocab.to_words(d_symbols[0]))
                print('[%d/%d][%d/%d] Loss: %.4f (val = %.4f)' % (epoch, args.epoch, i, len(train_dataset), rolling_loss.get(), val_loss))
            if args.max_iters and iteration >= args.max_iters:
                return rolling_loss.get()
    return rolling_loss.get()


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--data', type=str, default=None, help='Input sequence data', required=True)
    parser.add_argument('--vocab', type=str, default=None, help='Vocab data for input sequences', required=True)
    parser.add_argument('--weights', type=str, default=None, help='model path')
    parser.add_argument('--test', action='store_true', default=False, help='to test or not to test')
    parser.add_argument('--separate_decoder', action='store_true', default=False, help='use seperate weights for decoder')
    parser.add_argument('--epoch', type=int, default=5, help='number of epochs to perform')
    parser.add_argument('--batch_size', type=int, default=64, help='input batch size')
    parser.add_argument('--lr', type=float, default=0.001, help='learning rate')
    parser.add_argument('--beta1', type=float, default=0.9, help='beta1 for adam')
    parser.add_argument('--beta2', type=float, default=0.999, help='beta2 for adam')
    parser.add_argument('--seed', type=int, default=None, help='seed')
    parser.add_argument('--cuda', action='store_true', help='enables cuda')
    parser.add_argument('--workers', type=int, default=2, help='number of workers to help with loading/pre-processing data')
    parser.add_argument('--teacher_force', type=float, default=0.5, help='teacher force ratio')
    parser.add_argument('--max_seq_len', type=int, default=10, help='max sequence length')
    parser.add_argument('--out', type=str, default=None, help='output path for learnt models', required=True)
    # additions (the reference hard-codes 50 / 500 and has no iteration cap)
    parser.add_argument('--log_period', type=int, default=50)
    parser.add_argument('--save_model_period', type=int, default=500)
    parser.add_argument('--max_iters', type=int, default=0, help='stop after this many iterations (0 = run all epochs)')
    return parser


if __name__ == '__main__':
    main(build_parser().parse_args())
