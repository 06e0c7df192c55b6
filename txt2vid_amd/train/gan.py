"""CLI entry of the TGANv2 hot path — flag-for-flag txt2vid/train/gan.py:28-221 (+ trainer.py:15-42), so
`scripts/run_tganv2*.sh` / `config/*.json` drive this build. One process per GPU: launched under
`torch.distributed.run` it trains data-parallel over RCCL (the reference's `--ngpu` was never read).

    python -m txt2vid_amd.train.gan --data config/synth.json --G txt2vid.models.tganv2.gen.MultiScaleGen ...
"""
import argparse
import os

import torch

from .. import data
from ..gan.cond_gan import CondGan
from ..gan.losses import MixedGanLoss
from ..gan.trainer import add_params_to_parser, train, test
from ..optim import Adam, SGD
from ..util.log import status
from ..util.pick import load
from ..util.reflection import create_object
from ..util.torch.init import init
from .setup import setup


def _model(spec, args, kind='D', **kw):
    """`create_object(spec, cond_dim=...)` like the reference; convenience: a bare class name also gets
    `num_channels=--num_channels`, and a bare generator class `width = height = --frame_sizes[-1]` (the reference needs a
    JSON spec for anything but its 3-channel / 128x128 defaults)."""
    if isinstance(spec, str) and not os.path.exists(spec):
        kw.setdefault('num_channels', args.num_channels)
        if kind == 'G':                       # the reference's default generator is 128x128 whatever --frame_sizes says
            kw.setdefault('width', args.frame_sizes[-1])
            kw.setdefault('height', args.frame_sizes[-1])
    return create_object(spec, **kw)


def main(args):
    from .. import dist as tdist
    rank, world = tdist.init_from_env('nccl')
    from ..util.misc import limit_host_threads
    limit_host_threads()                 # an oversubscribed CPU thread pool starves the HIP runtime's submission thread
    if world > 1 and args.seed is None:
        import random
        args.seed = tdist.broadcast_seed(random.randint(1, 100000))      # one seed for all ranks: identical replicas
    seed, device = setup(args)
    status('%d cuda devices available; rank %d of %d' % (torch.cuda.device_count(), rank, world))
    vocab = load(args.vocab) if args.vocab else data.Vocab()
    txt_encoder = None
    if not args.dont_use_sent:
        if args.sent_weights:
            from ..util.reflection import alias_reference_modules
            alias_reference_modules()            # the reference pickles the whole Seq2Seq object under its own module path
            txt_encoder = torch.load(args.sent_weights, weights_only=False, map_location='cpu')
            if isinstance(txt_encoder, dict) and 'txt' in txt_encoder:
                txt_encoder = txt_encoder['txt']
            txt_encoder = txt_encoder.to(device)
            if hasattr(txt_encoder, 'differentiable'):
                txt_encoder.differentiable(False)      # a pre-training checkpoint may carry the autograd-path flag: the GAN loop
                #                                        detaches the sentence code (trainer.py:211-215), forward-only kernels
        else:
            txt_encoder = create_object(args.sent, vocab_size=len(vocab)).to(device)
            init(txt_encoder, init_method=args.sent_init_method or args.init_method)
    cond_dim = txt_encoder.encoder.encoding_size if txt_encoder is not None else 0
    gen = _model(args.G, args, kind='G', cond_dim=cond_dim)
    discrims = [_model(d, args, cond_dim=cond_dim) for d in args.D]
    init(gen, init_method=args.init_method)
    for d in discrims:
        init(d, init_method=args.init_method)
    gen.to(device)
    discrims = [d.to(device) for d in discrims]
    if args.M:
        raise NotImplementedError('--M sample mappings (TCWYT baseline) are outside the hot path')
    D_params = [{'params': d.parameters()} for d in discrims]
    G_params = [{'params': gen.parameters()}]
    if args.end2end and txt_encoder is not None:
        # train/gan.py:82-85: the text encoder trains with the GAN — its parameters join BOTH optimisers, the sentence code keeps
        # its graph (trainer.py:213-214), the D backward retains it (trainer.py:240) and the G backward runs through it again after
        # optD.step() has moved the encoder. That sequence only runs where the optimiser's in-place update does not bump autograd's
        # version counters: the reference's pinned torch 0.4.1 (updates through `.data`) — and here, where Adam is a kernel
        # writing through raw pointers. Stock torch >= 1.x raises "modified by an inplace operation" on the G backward.
        # Eager launches (the encoder graph spans the D and the G step: no graph replay), no gradient sink.
        D_params.append({'params': txt_encoder.parameters()})
        G_params.append({'params': txt_encoder.parameters()})
        txt_encoder.differentiable(True)
    if args.sgd:                                   # train/gan.py:86-89: momentum = beta1
        status('Using SGD')
        optD = SGD(D_params, lr=args.D_lr, momentum=args.D_beta1)
        optG = SGD(G_params, lr=args.G_lr, momentum=args.G_beta1)
    else:
        optD = Adam(D_params, lr=args.D_lr, betas=(args.D_beta1, args.D_beta2))
        optG = Adam(G_params, lr=args.G_lr, betas=(args.G_beta1, args.G_beta2))
    resumed_rng, resumed_iter = None, 0
    gan = CondGan(gen=gen, discrims=discrims, cond_encoder=txt_encoder, discrim_names=args.D_names,
                  discrim_lambdas=args.D_lambdas, gp_scale=float(world))
    if args.weights is not None:
        to_load = torch.load(args.weights, map_location=device, weights_only=False)
        gan.load_from_dict(to_load)
        if 'optD' in to_load:
            optD.load_state_dict(to_load['optD'])
        if 'optG' in to_load:
            optG.load_state_dict(to_load['optG'])
        resumed_rng, resumed_iter = to_load.get('rng_state'), int(to_load.get('iteration', 0))
        del to_load
    if world > 1:
        # replicas bit-identical before the first step (whatever each rank's init / checkpoint read produced), then per-rank
        # generators for everything drawn per iteration (z, sub-sample phases, GP alphas, caption permutations)
        from .. import functional as TF
        tdist.sync_replicas([gen] + list(discrims) + [txt_encoder])
        TF.bump_weight_epoch()
        import random
        import numpy as np
        for seeder in (random.seed, np.random.seed, torch.manual_seed):
            seeder(seed + rank + 1000003 * resumed_iter)     # (a resumed run does not replay the first run's draws)
    if resumed_rng is not None and rank == 0:
        # checkpoints written by this build carry the writer's (rank 0's) generator states: the resumed run continues the draw
        # sequence where the saved one stopped (reference-written files have no such key and start from --seed)
        from .setup import set_rng_state
        set_rng_state(resumed_rng)
    transform = data.default_transform(frame_size=[args.frame_sizes[-1]], num_channels=args.num_channels)
    dset = create_object(args.data, vocab=vocab, anno=args.anno, transform=transform, size=args.frame_sizes[-1],
                         channels=args.num_channels, seed=(args.seed or 0) + rank)
    dataset = data.get_loader(dset=dset, batch_size=args.batch_size, val=False, num_workers=args.workers, rank=rank,
                              world=world, seed=seed, device=device)
    status('GAN has %d parameters' % gan.count_params())
    if args.G_loss is None:
        args.G_loss = args.D_loss
    losses = MixedGanLoss(g_loss=create_object(args.G_loss), d_loss=create_object(args.D_loss))
    grad_sync = None
    if world > 1:
        from .. import functional as TF
        # --end2end: the text encoder sits in BOTH optimisers (train/gan.py:82-85), so its gradients travel with both exchanges
        shared = [txt_encoder] if (args.end2end and txt_encoder is not None) else []
        arenas = {'D': tdist.model_arena(list(discrims) + shared, TF.copy_into), 'G': tdist.model_arena([gen] + shared, TF.copy_into)}
        grad_sync = tdist.make_grad_sync(arenas, {'D': optD, 'G': optG}, world)
    if args.test:
        test(gan=gan, num_samples=args.num_samples, dataset=dataset, device=device, params=args,
             channel_first=not args.sequence_first, vocab=vocab)
    else:
        train(gan=gan, num_epoch=args.epochs, dataset=dataset, device=device, optD=optD, optG=optG, params=args,
              losses=losses, vocab=vocab, channel_first=not args.sequence_first, end2end=args.end2end,
              grad_sync=grad_sync, max_iters=args.max_iters)


# txt2vid/train/gan.py:163-221 — same flags, types and defaults; `max_iters` is new (stop after that many iterations)
FLAGS = """
test flag
num_samples int 1
seed int -
cuda flag
workers int 2
ngpu int 1
frame_sizes ints 64
num_channels int 1
random_frames int 0
opt_level str O2
epochs int 5
batch_size int 64
init_method str xavier
G_loss str -
G_lr float 0.0001
G_beta1 float 0.5
G_beta2 float 0.9
D_loss str txt2vid.gan.losses.VanillaGanLoss
D_lr float 0.0001
D_beta1 float 0.5
D_beta2 float 0.9
weights str -
sent_weights str -
data str - !
anno str -
vocab str -
M str -
G str - !
D strs - !
D_names strs -
D_lambdas floats -
sent str -
sent_init_method str -
dont_use_sent flag
end2end flag
sgd flag
sequence_first flag
debug flag
max_iters int -
"""


def build_parser():
    from ..util.cli import add_flags
    p = argparse.ArgumentParser()
    add_params_to_parser(p)
    return add_flags(p, FLAGS)


if __name__ == '__main__':
    main(build_parser().parse_args())
