"""Bi-LSTM sentence encoder / caption decoder — same surface / state_dict as txt2vid/models/txt/basic.py:4-101.
`nn.Embedding` / `nn.LSTM` / `nn.Linear` only HOLD the parameters (reference checkpoint layout); every forward runs on the HIP
kernels:
  * GAN loop (trainer.py:211-215 detaches the sentence code): `functional.lstm_encode`, forward only — embedding gather, one GEMM
    per layer and direction for the input projections of all time steps, one small launch per time step for the recurrence with
    packed-sequence masking;
  * text pre-training (train/txt.py:160-178), switched on with `Seq2Seq.differentiable(True)`: the same schedule through
    `functional.lstm_stack` (autograd Functions over `t2v_lstm_train_step[_bwd]`), `functional.embedding`, and the greedy /
    teacher-forced decoder `sample` (basic.py:73-101)."""
import torch
import torch.nn as nn


class RecurrentModel(nn.Module):
    # Class-level default: a Seq2Seq pickled by the reference (train/txt.py:185 saves the whole module) is rebuilt without
    # running this __init__, so the instance has no `with_grad` of its own; it then encodes on the forward-only kernels.
    with_grad = False

    def __init__(self, vocab_size=None, embed_size=256, hidden_size=256, encoding_size=256, num_layers=4, bi=True,
                 is_decoder=False):
        super().__init__()
        self.bi = bi
        self.num_layers = num_layers
        self.hidden_size = hidden_size // 2 if bi else hidden_size
        self.encoding_size = encoding_size
        self.vocab_size = vocab_size
        self.embed_size = embed_size
        self.embed = nn.Embedding(vocab_size, embed_size)
        self.lstm = nn.LSTM(embed_size, self.hidden_size, num_layers, batch_first=True, bidirectional=bi)
        self.is_decoder = is_decoder
        if is_decoder:
            self.to_vocab = nn.Linear(hidden_size, vocab_size)
        self.with_grad = False           # True: differentiable path (text pre-training)

    def _check_device(self):
        if not self.embed.weight.is_cuda:
            raise RuntimeError('the sentence encoder runs on the MI355X kernels only (no CPU path)')

    def _run_lstm(self, tokens, lengths, initial_state):
        """tokens [B,L] -> (out [B,L,D*H], (h list, c list)) on the differentiable path."""
        from ... import functional as TF
        B, L = int(tokens.shape[0]), int(lengths[0])
        dev = self.embed.weight.device
        len_dev = torch.tensor([int(l) for l in lengths], dtype=torch.int32).to(dev)
        x = TF.embedding(self.embed.weight, tokens[:, :L]).view(B, L, self.embed_size)
        return TF.lstm_stack(x, len_dev, self.lstm, self.hidden_size, self.num_layers, self.bi, initial_state)

    def forward(self, x, lengths=None, initial_state=None, raw_output=True):
        """tokens [B,L] (sorted by length, desc), lengths -> (out, hidden, hn[B, encoding])  (basic.py:49-70)."""
        from ... import functional as TF
        self._check_device()
        D = 2 if self.bi else 1
        # lengths as (L, int32 device tensor) is the graph-capturable form `GraphedSentenceEncoder` passes: forward-only kernels
        # whatever the flag says (the differentiable path reads the lengths on the host)
        device_lengths = isinstance(lengths, tuple) and len(lengths) == 2 and isinstance(lengths[1], torch.Tensor)
        if self.with_grad and torch.is_grad_enabled() and not device_lengths:
            out, (hs, cs) = self._run_lstm(x, lengths, initial_state)
            hidden = (hs, cs)                       # lists of [B,H], index layer * D + direction (differentiable)
            hn = TF.cat_features(hs[-2], hs[-1]) if self.bi else hs[-1].unsqueeze(0)
        else:
            if initial_state is not None:
                raise NotImplementedError('an initial state is only used by the decoder (differentiable path)')
            out, hidden = TF.lstm_encode(x, lengths, self.embed.weight, self.lstm, self.hidden_size, self.num_layers, self.bi)
            if self.bi:
                hn = hidden[0].view(self.num_layers, 2, -1, self.hidden_size)
                hn = TF.cat_features(hn[-1, 0], hn[-1, 1])
            else:
                hn = hidden[0].view(self.num_layers, 1, -1, self.hidden_size)[-1]
        if not raw_output:
            assert self.is_decoder
            B, L = out.shape[0], out.shape[1]
            out = TF.linear(out.reshape(B * L, D * self.hidden_size), self.to_vocab.weight, self.to_vocab.bias).view(B, L, -1)
            if L == 1:
                out = out.view(B, -1)
        return out, hidden, hn

    def sample(self, true_inputs=None, initial_hidden=None, max_seq_len=60, teacher_force=False):
        """Decoder roll-out (basic.py:73-101): start from `true_inputs[:, 0]`, one LSTM step per position from `initial_hidden`,
        logits through `to_vocab`, next input = arg-max (or `true_inputs[:, i]` when teacher forcing — the reference feeds the
        CURRENT position, kept). Returns (raw_outputs [B,T,V], symbols [B,T])."""
        from ... import functional as TF
        assert self.is_decoder and true_inputs is not None
        self._check_device()
        B = int(true_inputs.shape[0])
        D = 2 if self.bi else 1
        dev = self.embed.weight.device
        ones = torch.ones((B,), dtype=torch.int32).to(dev)
        if isinstance(initial_hidden[0], torch.Tensor):
            hidden = ([initial_hidden[0][i] for i in range(self.num_layers * D)], [initial_hidden[1][i] for i in range(self.num_layers * D)])
        else:
            hidden = initial_hidden
        inputs = true_inputs[:, 0]
        raw_outputs, symbols = [], []
        grad = self.with_grad and torch.is_grad_enabled()
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            for i in range(int(max_seq_len)):
                x = TF.embedding(self.embed.weight, inputs).view(B, 1, self.embed_size)
                out, hidden = TF.lstm_stack(x, ones, self.lstm, self.hidden_size, self.num_layers, self.bi, hidden)
                logits = TF.linear(out.view(B, D * self.hidden_size), self.to_vocab.weight, self.to_vocab.bias)
                predicted = TF.argmax_rows(logits)
                raw_outputs.append(logits)
                symbols.append(predicted)
                inputs = true_inputs[:, i] if teacher_force else predicted
        return TF.stack_steps(raw_outputs), torch.stack(symbols, 1)

    def create_initial_state(self):
        return torch.zeros(self.num_layers, 1, self.hidden_size)


class Seq2Seq(nn.Module):
    def __init__(self, separate_decoder=False, vocab_size=None):
        super().__init__()
        self.separate_decoder = separate_decoder
        self.encoder = RecurrentModel(vocab_size=vocab_size, is_decoder=not separate_decoder)
        self.decoder = RecurrentModel(vocab_size=vocab_size, is_decoder=True, bi=False) if separate_decoder else self.encoder

    def differentiable(self, on=True):
        """Switch the encoder / decoder to the autograd path (text pre-training); the GAN loop keeps the forward-only kernels."""
        self.encoder.with_grad = self.decoder.with_grad = bool(on)
        return self

    def encode(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.decoder.sample(*args, **kwargs)
