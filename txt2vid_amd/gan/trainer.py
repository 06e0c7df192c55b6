"""Training loop of the TGANv2 hot path — same entry points and flags as txt2vid/gan/trainer.py
(`add_params_to_parser`, `train`, `test`), with the loop body factored into `train_iteration` so that
bench.py and the parity tests drive exactly what `train()` runs.

Random draws follow the reference order on the HOST generators (SURVEY §7 "RNG parity"): 4x Subsample
phase (trainer.py:157-158), z (trainer.py:221, drawn on the CPU and copied), 3x Subsample in G,
gen_perm (numpy), 4x GP alpha (losses.py:140-145), gen_perm for real_pred (trainer.py:247).
"""
import os
import sys

import torch

from .. import functional as TF
from ..util.dir import ensure_exists
from ..util.log import status
from ..util.metrics import RollingAvg
from ..util.stopwatch import Stopwatch


def add_params_to_parser(parser):
    """Flag-for-flag txt2vid/gan/trainer.py:15-42 (including the `store_false` quirk of --no_mean_*)."""
    parser.add_argument('--data_is_imgs', action='store_true', default=False)
    parser.add_argument('--img_model', action='store_true', default=False)
    parser.add_argument('--log_period', type=int, default=20)
    parser.add_argument('--loss_window_size', type=int, default=20)
    parser.add_argument('--no_mean_discrim_loss', action='store_false', default=True)
    parser.add_argument('--no_mean_gen_loss', action='store_false', default=True)
    parser.add_argument('--no_graph', action='store_true', default=False, help='(new) eager launches instead of HIP-graph replay')
    parser.add_argument('--sample_batch_size', type=int, default=None)
    parser.add_argument('--discrim_steps', type=int, default=1)
    parser.add_argument('--gen_steps', type=int, default=1)
    parser.add_argument('--gp_lambda', type=float, default=-1)
    parser.add_argument('--save_initial', action='store_true', default=False)
    parser.add_argument('--save_initial_examples', action='store_true', default=False)
    parser.add_argument('--save_model_period', type=int, default=100)
    parser.add_argument('--save_example_period', type=int, default=100)
    parser.add_argument('--use_writer', action='store_true', default=False)
    parser.add_argument('--out', type=str, default='out')
    parser.add_argument('--out_samples', type=str, default='out_samples')
    parser.add_argument('--subsample_input', action='store_true', default=False)
    return parser


def multiscale_data(x, cond, frame_sizes, subsample_input=True):
    """Real-data pyramid — trainer.py:131-165. x: [B,C,T,H,W] on the device. Level i is the nearest
    resize of the (progressively batch/time sub-sampled) clip to frame_sizes[i]; one gather kernel per
    level does sub-sampling and resize in a single pass over the source. The Subsample phases come from
    `TF.draws` (host generator; device-resident under graph replay)."""
    n = len(frame_sizes)
    if n == 1:
        return [x], (None if cond is None else [cond])
    B, Cc, T, H, W = x.shape
    if subsample_input:
        t0s = TF.draws.multiscale_t0(n)
    else:
        t0s = [(0, None)] * n
    xs, conds = [], []
    sb, st = 1, 1                  # level tensor == x[::sb, :, t0::st]
    Bl, Tl = B, T
    for i in range(n):
        fs = frame_sizes[i] if i != n - 1 else None
        Ho, Wo = (fs, fs) if fs is not None else (H, W)
        t0, t0_dev = t0s[i]
        if sb == 1 and st == 1 and Ho == H and Wo == W:
            xs.append(x)
        else:
            xs.append(TF.pyramid_gather(x, Bl, Tl, Ho, Wo, sb, st, t0, t0_dev))
        if cond is not None:
            conds.append(cond)
        if subsample_input:
            if Tl % 2:
                raise ValueError('subsample_input needs a frame count divisible by 2**levels')
            st, sb = st * 2, sb * 2
            Bl, Tl = (Bl + 1) // 2, Tl // 2
            if cond is not None:
                # cond[::2] (trainer.py:160); --end2end keeps the sentence code's graph, so the slice must be differentiable
                cond = cond[::2] if cond.requires_grad else TF.stride_rows(cond, 2)
    return xs, (conds if conds else None)


class TrainStep(object):
    """The loop body trainer.py:199-267 after data loading, cut at the two points where data-parallel
    ranks exchange gradients:  part_d (pyramid, G forward, D loss + GP, D backward)  |sync D grads|
    part_g (Adam on D, real_pred, G loss, G backward)  |sync G grads|  part_end (Adam on G).
    Eager mode runs the three parts back to back; `GraphedTrainStep` captures each part as a HIP graph."""

    def __init__(self, gan, optD, optG, losses, params, device, end2end=False, grad_sync=None):
        if params.discrim_steps != 1 or params.gen_steps != 1:
            raise NotImplementedError('the canonical 1 D step : 1 G step schedule (scripts/run_tganv2*.sh) is built')
        self.gan, self.optD, self.optG, self.losses, self.params = gan, optD, optG, losses, params
        self.device, self.end2end, self.grad_sync = device, end2end, grad_sync
        self.lD = self.lG = self.fake = self.xs = self.conds = None
        self._one = None                  # dL/dL = 1 for both backward passes: one persistent 0-d tensor (no fill launch per pass)
        # weight / bias gradients land directly in flat per-model arenas (the ones the data-parallel exchange
        # all-reduces, when there is one): no per-parameter sums, no gather copies
        self.grad_sink = None
        if not end2end and torch.device(device).type == 'cuda':
            cached = gan.__dict__.get('_t2v_grad_sink')
            if cached is None or cached[0] is not grad_sync:
                from ..dist import model_arena
                arenas = list(grad_sync.arenas.values()) if grad_sync is not None else [
                    model_arena(list(gan.discrims), TF.copy_into), model_arena(gan.gen, TF.copy_into)]
                cached = (grad_sync, TF.GradSink(arenas))
                gan.__dict__['_t2v_grad_sink'] = cached          # one pair of arenas per model, reused by every step
            self.grad_sink = cached[1]

    def _root(self, loss):
        if self._one is None or self._one.device != loss.device or self._one.shape != loss.shape:
            self._one = TF.ones_like(loss.detach())
        return self._one

    def _arm_sink(self):
        TF.set_grad_sink(self.grad_sink)
        TF.grad_sink_reset()              # the step about to run clears this model's gradients first

    def part_d(self, x, cond):
        p = self.params
        self._arm_sink()
        self.xs, self.conds = multiscale_data(x, cond, p.frame_sizes, p.subsample_input)
        z = TF.draws.z(x.size(0), self.gan.gen.latent_size, self.device)       # CPU generator, then copy (parity)
        self.fake = self.gan(z, cond=self.conds[0] if self.conds is not None else None)
        loss = self.gan.discrim_step(real=self.xs, fake=[f.detach() for f in self.fake], cond=self.conds,
                                     loss=self.losses.discrim_loss, gp_lambda=p.gp_lambda)
        loss.backward(gradient=self._root(loss), retain_graph=self.end2end)
        TF.grad_sink_flush()              # the weight-gradient partial sums of the whole pass, summed in one launch
        self.lD = loss.detach()

    def part_g(self):
        self.optD.step()
        self._arm_sink()
        # D after its update, on the real batch (trainer.py:247). Its graph is only needed when the text
        # encoder trains end-to-end; otherwise nothing upstream of these predictions receives a gradient
        # that is ever used, so none is recorded (identical results, SURVEY §7 "wasted work").
        if not self.end2end and self.gan.can_fuse_gen_step(self.fake, self.xs):
            loss = self.gan.gen_step_fused(fake=self.fake, real=self.xs, cond=self.conds, loss=self.losses.gen_loss)
        else:
            if self.end2end:
                _, _, real_pred = self.gan.all_discrim_forward(real=self.xs, cond=self.conds, fake=None, loss=None)
            else:
                with torch.no_grad():
                    _, _, real_pred = self.gan.all_discrim_forward(real=self.xs, cond=self.conds, fake=None, loss=None)
            loss = self.gan.gen_step(fake=self.fake, real_pred=real_pred, cond=self.conds, loss=self.losses.gen_loss)
        loss.backward(gradient=self._root(loss))
        TF.grad_sink_flush()
        self.lG = loss.detach()

    def part_end(self):
        self.optG.step()

    def run(self, x, cond=None):
        self.part_d(x, cond)
        if self.grad_sync is not None:
            self.grad_sync('D')
        self.part_g()
        if self.grad_sync is not None:
            self.grad_sync('G')
        self.part_end()
        return self.lD, self.lG, self.fake, self.xs


def train_iteration(gan, x, cond, optD, optG, losses, params, device, end2end=False, z=None, grad_sync=None):
    """One eager pass of the loop body. x: [B,C,T,H,W] device tensor. Returns (lossD, lossG, fake, xs);
    the losses are 0-d device tensors (no host sync here)."""
    return TrainStep(gan, optD, optG, losses, params, device, end2end, grad_sync).run(x, cond)


_X_BY_MEMCPY = os.environ.get('T2V_DRAWS_MEMCPY') is not None
_DRAWS_LAST = os.environ.get('T2V_DRAWS_LAST') is not None


class GraphedTrainStep(object):
    """HIP-graph replay of the training iteration (unconditional path): after `warmup` eager iterations
    the three parts of `TrainStep` are captured once (shared memory pool) and every later iteration is
    3 graph launches + the eager gradient exchange between them (one process: ONE graph for the whole iteration). All per-iteration randomness is drawn
    on the host in the reference's order and uploaded into fixed buffers before the replay
    (`functional.StaticDraws`); the batch is copied into a fixed input buffer."""

    def __init__(self, gan, optD, optG, losses, params, device, batch_shape, grad_sync=None, warmup=3, cond_dim=0):
        self.ts = TrainStep(gan, optD, optG, losses, params, device, False, grad_sync)
        self.device, self.grad_sync, self.warmup = device, grad_sync, warmup
        n_levels = len(params.frame_sizes)
        self.draws = TF.StaticDraws(device, batch_shape[0], gan.gen.latent_size, n_levels, n_gen_phases=n_levels - 1,
                                    gp=params.gp_lambda > 0, subsample_input=params.subsample_input,
                                    n_perms=2 * len(gan.discrims) if cond_dim else 0)
        self.x = torch.empty(batch_shape, device=device, dtype=torch.float32)
        # conditional path: the sentence code is computed outside the graphs (its sequence length varies) into a fixed buffer
        self.cond = torch.empty((batch_shape[0], cond_dim), device=device, dtype=torch.float32) if cond_dim else None
        self.graphs = None
        self.n_graphs = 0                    # graphs per iteration of the last capture (1 without a gradient exchange, else 3)
        self.n = 0
        # warm-up and capture run on ONE side stream (the autograd graph's AccumulateGrad nodes remember
        # the stream they were created on; capture must see the same one)
        self.side = torch.cuda.Stream(device=device)

    def _eager(self):
        self.ts.run(self.x, self.cond)

    def _capture(self):
        for opt in (self.ts.optD, self.ts.optG):
            opt.make_capturable(self.device)
        # drop every reference into the previous iteration's autograd graph before capturing
        self.ts.lD = self.ts.lG = self.ts.fake = self.ts.xs = self.ts.conds = None
        torch.cuda.synchronize()
        g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.draws.rewind()
        gs = self.grad_sync
        # thread-local capture mode: the data loader's pin-memory thread keeps issuing pinned allocations and H2D copies
        # while this thread captures; in the default (global) mode any such call invalidates the capture
        mode = 'thread_local'
        if gs is None and os.environ.get('T2V_THREE_GRAPHS') is None:      # (T2V_THREE_GRAPHS: developer A/B switch)
            # one process, no gradient exchange between the parts: the whole iteration is ONE graph (two replay launches fewer)
            with torch.cuda.graph(g1, stream=self.side, capture_error_mode=mode):
                self.ts.part_d(self.x, self.cond)
                self.ts.part_g()
                self.ts.part_end()
            torch.cuda.synchronize()
            self.graphs = (g1,)
            self.n_graphs = 1
            if self.ts.grad_sink is not None:
                self.ts.grad_sink.frozen += 1
            return
        with torch.cuda.graph(g1, stream=self.side, capture_error_mode=mode):
            self.ts.part_d(self.x, self.cond)
            if gs is not None:
                gs.pre('D')                 # p.grad -> arena: captured (the replayed backward rewrites the same buffers)
        if gs is not None:
            gs.exchange('D')
            gs.post('D')                    # Adam (captured next) reads the arena views
        with torch.cuda.graph(g2, pool=g1.pool(), stream=self.side, capture_error_mode=mode):
            self.ts.part_g()
            if gs is not None:
                gs.pre('G')
        if gs is not None:
            gs.exchange('G')
            gs.post('G')
        with torch.cuda.graph(g3, pool=g1.pool(), stream=self.side, capture_error_mode=mode):
            self.ts.part_end()
        torch.cuda.synchronize()
        self.graphs = (g1, g2, g3)
        self.n_graphs = 3
        if self.ts.grad_sink is not None:
            self.ts.grad_sink.frozen += 1        # the graphs bake in the sink's slab addresses and table slots

    def release(self):
        """Drop the captured graphs (a new batch shape is about to be captured): the gradient sink may grow / recycle again."""
        if self.graphs is not None:
            torch.cuda.synchronize()
            self.graphs = None
            if self.ts.grad_sink is not None:
                self.ts.grad_sink.frozen = max(0, self.ts.grad_sink.frozen - 1)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def step(self, x, cond=None):
        """x: [B,C,T,H,W] device (or pinned host) tensor; cond: [B,cond_dim] sentence codes (conditional path). Returns
        (lossD, lossG) 0-d device tensors that are overwritten by the next call."""
        old = TF.draws
        TF.set_draws(self.draws)
        try:
            if not _DRAWS_LAST:
                self.draws.begin_step()                  # (host RNG work first: the launches below then reach the queue back to back)
            if x.is_cuda and x.dtype == torch.float32 and x.shape == self.x.shape and not _X_BY_MEMCPY:
                TF.copy_into(x, self.x)                  # (a copy kernel: stays on the compute queue in front of the graph, see StaticDraws.begin_step)
            else:
                self.x.copy_(x, non_blocking=True)
            if self.cond is not None:
                TF.copy_into(cond.detach(), self.cond)
            if _DRAWS_LAST:
                self.draws.begin_step()
            if self.graphs is None and self.n >= self.warmup:
                self._capture()          # the capture pass also executes nothing: replay right after
            if self.graphs is None:
                cur = torch.cuda.current_stream(self.device)
                self.side.wait_stream(cur)
                with torch.cuda.stream(self.side):
                    self._eager()
                cur.wait_stream(self.side)
            else:
                self.graphs[0].replay()
                if len(self.graphs) == 3:                 # data-parallel: the two exchanges sit between the captured parts
                    if self.grad_sync is not None:
                        self.grad_sync.exchange('D')
                    self.graphs[1].replay()
                    if self.grad_sync is not None:
                        self.grad_sync.exchange('G')
                    self.graphs[2].replay()
            self.n += 1
        finally:
            TF.set_draws(old)
        return self.ts.lD, self.ts.lG


class GraphedSentenceEncoder(object):
    """HIP-graph replay of the sentence encoder's forward (`cond_encoder.encode`, ~90 small launches: embedding gather, one GEMM
    per layer and direction, one launch per time step): one graph per sequence length — caption batches differ in their longest
    caption, and the per-sample lengths are read on the device. The first batch of a new length runs eagerly (it also fills the
    packed / transposed weight caches), the second is captured, later ones replay. Returns the sentence codes [B, encoding] in a
    buffer that the next call of the same length overwrites."""

    def __init__(self, cond_encoder, device):
        self.enc, self.device = cond_encoder, device
        self.entries = {}                # (B, L) -> [tokens int32 [B,L], lengths int32 [B], graph | None, codes | None, last lengths]
        self.side = torch.cuda.Stream(device=device)
        self.ring = [torch.zeros((1024,), dtype=torch.int32).pin_memory() for _ in range(8)]
        self.ring_pos = 0

    def encode(self, tokens, lengths):
        B, L = int(tokens.shape[0]), int(lengths[0])
        if B > 1024:
            return self.enc.encode(tokens, lengths)[2].detach()
        ent = self.entries.get((B, L))
        if ent is None:
            ent = [torch.zeros((B, L), dtype=torch.int32).to(self.device), torch.zeros((B,), dtype=torch.int32).to(self.device),
                   None, None]
            self.entries[(B, L)] = ent
            return self.enc.encode(tokens, lengths)[2].detach()
        tok, len_dev, graph, codes = ent[:4]
        tok.copy_(tokens[:, :L], non_blocking=True)                         # int64 -> int32, device or (pinned) host source
        lens = [int(n) for n in lengths]
        if len(ent) < 5 or ent[4] != lens:
            # upload through a small ring of pinned buffers: a pageable source would make the host wait for the stream (and
            # with it for the previous iteration) before it can go on to launch this one
            slot = self.ring[self.ring_pos % len(self.ring)]
            self.ring_pos += 1
            slot[:B].copy_(torch.tensor(lens, dtype=torch.int32))
            len_dev.copy_(slot[:B], non_blocking=True)
            if len(ent) < 5:
                ent.append(lens)
            else:
                ent[4] = lens
        if graph is None:
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=self.side, capture_error_mode='thread_local'):
                codes = self.enc.encode(tok, (L, len_dev))[2]
            ent[2], ent[3] = graph, codes
        graph.replay()
        return codes


def test(gan=None, num_samples=1, dataset=None, device=None, params=None, channel_first=True, vocab=None):
    """Sampling path — trainer.py:44-90. `num_samples` times: take the FIRST batch of the loader (the reference `break`s after
    one), encode its captions, draw z, run the eval-mode generator (one full [B,C,16,S,S] clip per latent: no sub-sampling,
    last level only unless `output_blocks` asks for more) and write, with the reference's file names,
    `real_<i>.png`, `sentences_<i>_<j>.txt` and one `<h>x<w>_<i>_<j>.jpg` grid per rendered level."""
    from .samples import save_frames, save_sentences
    from ..data import DevicePrefetcher
    ensure_exists(params.out_samples)
    gan.gen.eval()
    with torch.no_grad():
        for i in range(num_samples):
            pre = DevicePrefetcher(dataset, device)
            j = 0
            x, y = pre.next()
            if x is None:
                break
            x = TF.video_to_channel_first(x)
            cond = None
            if gan.cond_encoder is not None and len(y) >= 2:
                _, _, cond = gan.cond_encoder.encode(y[0], y[1])
            z = torch.randn(x.size(0), gan.gen.latent_size).to(device)
            fake = gan(z, cond=cond)
            save_frames(x, '%s/real_%d.png' % (params.out_samples, i), is_images=getattr(params, 'img_model', False))
            if cond is not None and vocab is not None:
                save_sentences(y[0], path='%s/sentences_%d_%d.txt' % (params.out_samples, i, j), vocab=vocab)
            for f in fake:
                h, w = f.size(3), f.size(4)
                path = '%s/%dx%d_%d_%d.jpg' % (params.out_samples, h, w, i, j)
                status('saving to %s' % path)
                save_frames(f, path, is_images=getattr(params, 'img_model', False))
    gan.gen.train()


def train(gan=None, num_epoch=None, dataset=None, device=None, optD=None, optG=None, params=None, vocab=None, losses=None,
          channel_first=True, end2end=True, grad_sync=None, max_iters=None):
    """txt2vid/gan/trainer.py:111-333. `dataset` yields (videos [B,T,C,H,W], tokens, lengths) batches
    (host tensors); a pinned-memory prefetch thread overlaps H2D with the step."""
    assert channel_first
    if params.sample_batch_size is None:
        params.sample_batch_size = params.batch_size
    ensure_exists(params.out)
    ensure_exists(params.out_samples)
    gen_loss = RollingAvg(window_size=params.loss_window_size)
    discrim_loss = RollingAvg(window_size=params.loss_window_size)
    avg_iter = RollingAvg(window_size=max(1, params.log_period))
    avg_load = RollingAvg(window_size=max(1, params.log_period))
    load_watch, iter_watch = Stopwatch(), Stopwatch()
    from ..data import DevicePrefetcher
    from ..dist import rank_world
    writer = rank_world()[0] == 0            # data-parallel replicas are identical: rank 0 alone writes checkpoints and samples
    iteration = 0
    # HIP-graph replay of the iteration (3 graphs, see GraphedTrainStep) unless --no_graph / --end2end: the eager loop is
    # host-bound (~1 000 launches at ~20 us of Python each)
    use_graph = not end2end and not getattr(params, 'no_graph', False) and torch.device(device).type == 'cuda'
    graphed, graphed_key, pending, loss_ring, graphed_encoder = None, None, None, None, None
    for epoch in range(num_epoch):
        if params.log_period > 0:
            status('Epoch %d started' % (epoch + 1))
        load_watch.start()
        iter_watch.start()
        if hasattr(getattr(dataset, 'sampler', None), 'set_epoch'):
            dataset.sampler.set_epoch(epoch)                 # a new permutation per epoch on every rank
        pre = DevicePrefetcher(dataset, device)
        i = 0
        x, y = pre.next()
        while x is not None:
            iteration = epoch * len(dataset) + i + 1
            load_watch.stop()
            avg_load.update(load_watch.elapsed_time)
            x = TF.video_to_channel_first(x)                      # [B,T,C,H,W] -> [B,C,T,H,W] (trainer.py:204)
            cond = None
            if gan.cond_encoder is not None and len(y) >= 2:
                if use_graph:
                    if graphed_encoder is None:
                        graphed_encoder = GraphedSentenceEncoder(gan.cond_encoder, device)
                    cond = graphed_encoder.encode(y[0], y[1])
                else:
                    _, _, cond = gan.cond_encoder.encode(y[0], y[1])
                    if not end2end:
                        cond = cond.detach()
            if use_graph:
                key = (tuple(x.shape), None if cond is None else tuple(cond.shape))
                if graphed is None or key != graphed_key:          # first batch (or a new batch shape): capture again
                    if graphed is not None:
                        graphed.release()
                    graphed = GraphedTrainStep(gan, optD, optG, losses, params, device, tuple(x.shape), grad_sync=grad_sync,
                                               warmup=2, cond_dim=0 if cond is None else cond.shape[1])
                    graphed_key = key
                    if params.log_period > 0:
                        status('HIP-graph replay of the iteration for batches of shape %s (2 eager iterations, then capture)' % (key[0],))
                lD, lG = graphed.step(x, cond)
                fake, xs = graphed.ts.fake, graphed.ts.xs
                # Read the losses ONE iteration late, through a pinned buffer and an event: the host goes on to prepare
                # and launch the next iteration while this one runs (the replay overwrites its loss scalars, hence the
                # copies). The rolling averages lag by one iteration.
                if loss_ring is None:
                    loss_ring = [(torch.empty(2, device=lD.device), torch.empty(2).pin_memory(), torch.cuda.Event()) for _ in range(2)]
                dbuf, hbuf, ev = loss_ring[iteration & 1]
                if _X_BY_MEMCPY:
                    TF.copy_into(lD.reshape(1), dbuf[0:1])
                    TF.copy_into(lG.reshape(1), dbuf[1:2])
                    hbuf.copy_(dbuf, non_blocking=True)
                else:                                    # copy kernels writing the pinned buffer in place: no copy-engine hop behind the graph
                    TF._copy2d(lD.reshape(1), 0, 1, hbuf, 0, 1, 1, 1)
                    TF._copy2d(lG.reshape(1), 0, 1, hbuf, 1, 1, 1, 1)
                ev.record()
                if pending is not None:
                    pending[1].synchronize()
                    discrim_loss.update(float(pending[0][0]))
                    gen_loss.update(float(pending[0][1]))
                pending = (hbuf, ev)
            else:
                lD, lG, fake, xs = train_iteration(gan, x, cond, optD, optG, losses, params, device, end2end=end2end,
                                                   grad_sync=grad_sync)
                discrim_loss.update(float(lD))
                gen_loss.update(float(lG))
            # checkpoint: the reference tests `save_example_period` here (trainer.py:269) and never reads
            # --save_model_period; the intended flag is used (SURVEY §8a defect 3).
            if writer and ((iteration == 1 and params.save_initial) or (params.save_model_period > 0 and
                                                                        iteration % params.save_model_period == 0)):
                from ..train.setup import get_rng_state
                # extra keys next to the reference's layout (its loader ignores them; its files, which lack them, still load here)
                to_save = {'optG': optG.state_dict(), 'optD': optD.state_dict(), 'iteration': iteration, 'rng_state': get_rng_state()}
                to_save.update(gan.save_dict())
                torch.save(to_save, '%s/iter_%d_lossG_%.4f_lossD_%.4f' % (params.out, iteration, gen_loss.get(),
                                                                          discrim_loss.get()))
            if params.log_period > 0 and iteration % params.log_period == 0:
                sys.stdout.flush()
                status('[%d/%d; %d/%d] - Iter %d, Loss_D: %.4f Loss_G: %.4f (%.2fGB used; %.2fGB cached) - %.4f sec/iter; '
                       '%.4f sec/batch load' % (epoch, num_epoch, i, len(dataset), iteration, discrim_loss.get(),
                                                gen_loss.get(), torch.cuda.max_memory_allocated() / 1e9,
                                                torch.cuda.max_memory_reserved() / 1e9, avg_iter.get(), avg_load.get()))
                torch.cuda.reset_peak_memory_stats()
            if writer and params.save_example_period > 0 and ((iteration == 1 and params.save_initial_examples) or
                                                              iteration % params.save_example_period == 0):
                from .samples import save_frames, save_sentences
                status('saving to %s (iteration %d)' % (params.out_samples, iteration))
                save_frames(xs[0], '%s/real_samples.png' % params.out_samples)
                for f in fake:
                    h, w = f.size(3), f.size(4)
                    save_frames(f.detach(), '%s/fake_samples_epoch_%03d_iter_%06d_%dx%d.png' % (params.out_samples, epoch,
                                                                                                 iteration, h, w))
                if cond is not None and vocab is not None:
                    save_sentences(y[0], path='%s/sentences_epoch%03d_iter_%06d.txt' % (params.out_samples, epoch, iteration),
                                   vocab=vocab)
            load_watch.start()
            iter_watch.stop()
            avg_iter.update(iter_watch.elapsed_time)
            iter_watch.start()
            x, y = pre.next()
            i += 1
            if max_iters is not None and iteration >= max_iters:
                return iteration
    return iteration
