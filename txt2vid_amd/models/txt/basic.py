"""Bi-LSTM sentence encoder — same surface / state_dict as txt2vid/models/txt/basic.py:4-70 (encode path).
`nn.Embedding` / `nn.LSTM` only HOLD the parameters (reference checkpoint layout); the forward runs on the HIP kernels
(`functional.lstm_encode`: embedding gather, one GEMM per layer and direction for the input projections of all time steps,
one small launch per time step for the recurrence with packed-sequence masking). Forward only: the GAN loop detaches the
sentence code (trainer.py:211-215) unless --end2end, which is not built. The decoder / `sample` (pre-training) is out of scope."""
import torch
import torch.nn as nn


class RecurrentModel(nn.Module):
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
            self.to_vocab = nn.Linear(hidden_size, vocab_size)      # kept for checkpoint compatibility

    def forward(self, x, lengths=None, initial_state=None, raw_output=True):
        """tokens [B,L] (sorted by length, desc), lengths -> (out, hidden, hn[B, encoding])  (basic.py:49-70)."""
        from ... import functional as TF
        if initial_state is not None:
            raise NotImplementedError('an initial state is only used by the decoder (pre-training), outside the hot path')
        if not self.embed.weight.is_cuda:
            raise RuntimeError('the sentence encoder runs on the MI355X kernels only (no CPU path)')
        out, hidden = TF.lstm_encode(x, lengths, self.embed.weight, self.lstm, self.hidden_size, self.num_layers, self.bi)
        if self.bi:
            hn = hidden[0].view(self.num_layers, 2, -1, self.hidden_size)
            hn = TF.cat_features(hn[-1, 0], hn[-1, 1])
        else:
            hn = hidden[0].view(self.num_layers, 1, -1, self.hidden_size)[-1]
        if not raw_output:
            raise NotImplementedError('decoder output head is outside the hot path')
        return out, hidden, hn


class Seq2Seq(nn.Module):
    def __init__(self, separate_decoder=False, vocab_size=None):
        super().__init__()
        self.separate_decoder = separate_decoder
        self.encoder = RecurrentModel(vocab_size=vocab_size, is_decoder=not separate_decoder)
        self.decoder = RecurrentModel(vocab_size=vocab_size, is_decoder=True, bi=False) if separate_decoder else self.encoder

    def encode(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)

    def decode(self, *args, **kwargs):
        raise NotImplementedError('caption decoding (text pre-training, train/txt.py) is outside the hot path')
