"""Training loop of the TGANv2 hot path — same entry points and flags as txt2vid/gan/trainer.py
(`add_params_to_parser`, `train`, `test`), with the loop body factored into `train_iteration` so that
bench.py and the parity tests drive exactly what `train()` runs.

Random draws follow the reference order on the HOST generators (SURVEY §7 "RNG parity"): 4x Subsample
phase (trainer.py:157-158), z (trainer.py:221, drawn on the CPU and copied), 3x Subsample in G,
gen_perm (numpy), 4x GP alpha (losses.py:140-145), gen_perm for real_pred (trainer.py:247).
"""
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
    level does sub-sampling and resize in a single pass over the source."""
    n = len(frame_sizes)
    if n == 1:
        return [x], (None if cond is None else [cond])
    B, Cc, T, H, W = x.shape
    xs, conds = [], []
    sb, st, t0 = 1, 1, 0           # level tensor == x[::sb, :, t0::st]
    Bl, Tl = B, T
    for i in range(n):
        fs = frame_sizes[i] if i != n - 1 else None
        Ho, Wo = (fs, fs) if fs is not None else (H, W)
        if sb == 1 and st == 1 and Ho == H and Wo == W:
            xs.append(x)
        else:
            xs.append(TF.pyramid_gather(x, Bl, Tl, Ho, Wo, sb, st, t0))
        if cond is not None:
            conds.append(cond)
        if subsample_input:
            bt = int(torch.randint(2, (1,)))                 # Subsample.forward draw (layers.py:108)
            t0, st, sb = t0 + bt * st, st * 2, sb * 2
            Bl, Tl = (Bl + 1) // 2, (Tl - bt + 1) // 2
            if cond is not None:
                cond = TF.stride_rows(cond, 2)
    return xs, (conds if conds else None)


def train_iteration(gan, x, cond, optD, optG, losses, params, device, end2end=False, z=None, grad_sync=None):
    """One pass of the loop body trainer.py:199-267 after data loading. x: [B,C,T,H,W] device tensor.
    Returns (lossD, lossG) as 0-d device tensors (no host sync here)."""
    batch_size = x.size(0)
    xs, conds = multiscale_data(x, cond, params.frame_sizes, params.subsample_input)
    if z is None:
        z = torch.randn(batch_size, gan.gen.latent_size)          # CPU generator, then copy (parity)
    z = z.to(device, non_blocking=True)
    fake = gan(z, cond=conds[0] if conds is not None else None)

    total_d = None
    for j in range(params.discrim_steps):
        loss = gan.discrim_step(real=xs, fake=[f.detach() for f in fake], cond=conds, loss=losses.discrim_loss,
                                gp_lambda=params.gp_lambda)
        if not params.no_mean_discrim_loss:
            loss = TF.scalar_sum([loss], [1.0 / params.discrim_steps])
        loss.backward(retain_graph=(j != params.discrim_steps - 1) or end2end)
        if grad_sync is not None:
            grad_sync('D')
        optD.step()
        total_d = loss.detach() if total_d is None else TF.scalar_sum([total_d, loss.detach()])

    # D after its update, on the real batch (trainer.py:247). Its graph is only needed when the text
    # encoder trains end-to-end; otherwise nothing upstream of these predictions receives a gradient
    # that is ever used, so no graph is recorded (identical results, SURVEY §7 "wasted work").
    if end2end:
        _, _, real_pred = gan.all_discrim_forward(real=xs, cond=conds, fake=None, loss=None)
    else:
        with torch.no_grad():
            _, _, real_pred = gan.all_discrim_forward(real=xs, cond=conds, fake=None, loss=None)

    total_g = None
    for j in range(params.gen_steps):
        if j != 0:
            fake = gan(z, cond=conds[0] if conds is not None else None)
        loss = gan.gen_step(fake=fake, real_pred=real_pred, cond=conds, loss=losses.gen_loss)
        if not params.no_mean_gen_loss:
            loss = TF.scalar_sum([loss], [1.0 / params.gen_steps])
        loss.backward(retain_graph=j != params.gen_steps - 1)
        if grad_sync is not None:
            grad_sync('G')
        optG.step()
        total_g = loss.detach() if total_g is None else TF.scalar_sum([total_g, loss.detach()])
    return total_d, total_g, fake, xs


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
    iteration = 0
    for epoch in range(num_epoch):
        if params.log_period > 0:
            status('Epoch %d started' % (epoch + 1))
        load_watch.start()
        iter_watch.start()
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
                _, _, cond = gan.cond_encoder.encode(y[0], y[1])
                if not end2end:
                    cond = cond.detach()
            lD, lG, fake, xs = train_iteration(gan, x, cond, optD, optG, losses, params, device, end2end=end2end,
                                               grad_sync=grad_sync)
            discrim_loss.update(float(lD))
            gen_loss.update(float(lG))
            # checkpoint: the reference tests `save_example_period` here (trainer.py:269) and never reads
            # --save_model_period; the intended flag is used (SURVEY §8a defect 3).
            if (iteration == 1 and params.save_initial) or (params.save_model_period > 0 and
                                                            iteration % params.save_model_period == 0):
                to_save = {'optG': optG.state_dict(), 'optD': optD.state_dict(), 'iteration': iteration}
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
            if params.save_example_period > 0 and ((iteration == 1 and params.save_initial_examples) or
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
