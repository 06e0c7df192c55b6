"""Timing of one text pre-training iteration (train/txt.py loop body: encode -> teacher-forced decode -> cross entropy -> backward ->
Adam) on the HIP path, next to the CPU oracle's explicit-loop restatement on the host cores.
  python tools/txt_bench.py [batch=64] [length=10] [vocab=3000] [iters=20]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from txt2vid_amd.models.txt.basic import Seq2Seq  # noqa: E402
from txt2vid_amd.optim import Adam  # noqa: E402
from txt2vid_amd.train.txt import pretrain_loss  # noqa: E402
from txt2vid_amd.util.misc import host_threads, limit_host_threads  # noqa: E402
from txt2vid_amd.util.torch.init import init  # noqa: E402


def main():
    B, L, V, iters = [int(a) for a in sys.argv[1:5]] + [64, 10, 3000, 20][len(sys.argv) - 1:]
    limit_host_threads()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    m = Seq2Seq(vocab_size=V)
    init(m, 'xavier')
    m.to(dev).differentiable(True)
    opt = Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
    lengths = sorted([max(2, L - (i * L) // (2 * B)) for i in range(B)], reverse=True)
    sent = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lengths):
        sent[b, :n] = torch.randint(4, V, (n,))
    sent = sent.to(dev)

    def step():
        m.zero_grad()
        loss, _ = pretrain_loss(m, sent, lengths, True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    res = {'workload': 'text auto-encoder pre-training iteration, B=%d, L=%d (ragged %d..%d), V=%d, 4-layer Bi-LSTM 2x128, Adam' % (B, L, lengths[0], lengths[-1], V),
           'hip_ms_per_iter': dt * 1e3, 'sentences_per_s': B / dt, 'loss': float(loss)}
    # the CPU oracle (explicit loops, autograd on the host)
    from oracle import tganv2_oracle as O
    from oracle import txt_oracle as TO
    torch.set_num_threads(host_threads())
    P = {k: O.recipe_tensor(k, s).requires_grad_(True) for k, s in TO.seq2seq_shapes(V).items()}
    sc = sent.cpu()
    t0 = time.perf_counter()
    n = 0
    while n < 3:
        l = TO.pretrain_loss(P, sc, lengths, True)[0]
        l.backward()
        n += 1
    res['cpu_oracle_ms_per_iter'] = (time.perf_counter() - t0) / n * 1e3
    res['cpu_threads'] = host_threads()
    print(json.dumps(res))


if __name__ == '__main__':
    main()
