"""CPU oracle for the text auto-encoder pre-training step (TEST INFRASTRUCTURE — never shipped, never imported by the product
package `txt2vid_amd/`; only `tests/` may import this file).

A restatement, in plain fp32 PyTorch CPU ops with EXPLICIT time loops (no `nn.LSTM`, no `pack_padded_sequence`), of what the
reference computes in one iteration of `txt2vid/train/txt.py:160-178`:
`Seq2Seq.encode` (`models/txt/basic.py:49-70`: embedding, packed 4-layer Bi-LSTM, `hn = cat(h_fwd[-1], h_bwd[-1])`),
`Seq2Seq.decode` = `RecurrentModel.sample` (`basic.py:73-101`: one LSTM step per position from the encoder's state, `to_vocab`,
arg-max, teacher forcing feeding the CURRENT position), `nn.CrossEntropyLoss` over `decoded.permute(0, 2, 1)` (`txt.py:158,172`).
Functions take a flat ``dict[str, Tensor]`` keyed like `Seq2Seq.state_dict()` (`encoder.*`; without `--separate_decoder` the
decoder IS the encoder module).

Parity pin: `tests/golden/txt_pretrain.npz`, recorded from the real reference by `tests/golden/make_golden.py txt_pretrain`;
`tests/test_oracle_golden.py::test_txt_pretrain` checks this file against it.
"""
import torch


def seq2seq_shapes(vocab_size, embed=256, hidden=256, layers=4):
    """Keys / shapes of `Seq2Seq(vocab_size).state_dict()` under the `encoder.` prefix (basic.py:25-47)."""
    H = hidden // 2
    sh = {'encoder.embed.weight': (vocab_size, embed)}
    for l in range(layers):
        for sfx in ('', '_reverse'):
            sh['encoder.lstm.weight_ih_l%d%s' % (l, sfx)] = (4 * H, embed if l == 0 else 2 * H)
            sh['encoder.lstm.weight_hh_l%d%s' % (l, sfx)] = (4 * H, H)
            sh['encoder.lstm.bias_ih_l%d%s' % (l, sfx)] = (4 * H,)
            sh['encoder.lstm.bias_hh_l%d%s' % (l, sfx)] = (4 * H,)
    sh['encoder.to_vocab.weight'] = (vocab_size, hidden)
    sh['encoder.to_vocab.bias'] = (vocab_size,)
    return sh


def _cell(x_t, h, c, P, sfx):
    """One nn.LSTM cell step, gate order i, f, g, o."""
    pre = x_t @ P['encoder.lstm.weight_ih' + sfx].t() + P['encoder.lstm.bias_ih' + sfx] \
        + h @ P['encoder.lstm.weight_hh' + sfx].t() + P['encoder.lstm.bias_hh' + sfx]
    i, f, g, o = pre.chunk(4, 1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def lstm_packed(x, lengths, P, layers=4, init=None):
    """Bi-LSTM over a padded batch with `pack_padded_sequence` semantics (basic.py:53-56): sample b only advances while
    t < lengths[b]; its outputs beyond are zero; the reverse direction starts at each sample's own last token.
    x [B,L,In] -> out [B,L,2H], (h_n, c_n) [layers*2, B, H]."""
    B, L, _ = x.shape
    lens = torch.tensor([int(v) for v in lengths])
    hs, cs = [], []
    inp = x
    for l in range(layers):
        outs = []
        for d, sfx in enumerate(('_l%d' % l, '_l%d_reverse' % l)):
            H = P['encoder.lstm.weight_hh' + sfx].shape[1]
            h = init[0][l * 2 + d] if init is not None else x.new_zeros(B, H)
            c = init[1][l * 2 + d] if init is not None else x.new_zeros(B, H)
            out_t = [None] * L
            for t in (range(L - 1, -1, -1) if d else range(L)):
                act = (t < lens).to(x.dtype).unsqueeze(1)
                h2, c2 = _cell(inp[:, t], h, c, P, sfx)
                h = act * h2 + (1 - act) * h
                c = act * c2 + (1 - act) * c
                out_t[t] = act * h2
            outs.append(torch.stack(out_t, 1))
            hs.append(h)
            cs.append(c)
        inp = torch.cat(outs, 2)
    return inp, (torch.stack(hs, 0), torch.stack(cs, 0))


def encode(P, tokens, lengths, layers=4):
    """`RecurrentModel.forward` (basic.py:49-70) -> (out, (h_n, c_n), hn [B, 2H])."""
    L = int(lengths[0])
    x = P['encoder.embed.weight'][tokens[:, :L]]
    out, (h_n, c_n) = lstm_packed(x, lengths, P, layers)
    return out, (h_n, c_n), torch.cat((h_n[-2], h_n[-1]), 1)


def sample(P, true_inputs, hidden, max_seq_len, teacher_force, layers=4):
    """`RecurrentModel.sample` (basic.py:73-101): every sample takes every step (no packing), the Bi-LSTM sees a length-1
    sequence per step, so both directions simply advance their own state."""
    B = true_inputs.shape[0]
    inputs = true_inputs[:, 0]
    raw, syms = [], []
    ones = [1] * B
    for i in range(int(max_seq_len)):
        x = P['encoder.embed.weight'][inputs].unsqueeze(1)
        out, hidden = lstm_packed(x, ones, P, layers, init=hidden)
        logits = out[:, 0] @ P['encoder.to_vocab.weight'].t() + P['encoder.to_vocab.bias']
        pred = logits.max(1)[1]
        raw.append(logits)
        syms.append(pred)
        inputs = true_inputs[:, i] if teacher_force else pred
    return torch.stack(raw, 1), torch.stack(syms, 1)


def pretrain_loss(P, tokens, lengths, teacher_force, reduction='mean'):
    """The loss of one pre-training iteration (txt.py:160-172). Padding positions hold token 0 and COUNT as class-0 targets
    (the reference's `pad_packed_sequence` round trip of the token matrix zero-fills them). Returns (loss, decoded, symbols, hn)."""
    L = int(lengths[0])
    _, hidden, hn = encode(P, tokens, lengths)
    lens = torch.tensor([int(v) for v in lengths]).unsqueeze(1)
    targets = torch.where(torch.arange(L).unsqueeze(0) < lens, tokens[:, :L], torch.zeros_like(tokens[:, :L]))
    decoded, symbols = sample(P, tokens, hidden, L, teacher_force)
    logp = torch.log_softmax(decoded, 2)
    nll = -logp.gather(2, targets.unsqueeze(2)).squeeze(2)
    loss = nll.mean() if reduction == 'mean' else nll.sum()
    return loss, decoded, symbols, hn
