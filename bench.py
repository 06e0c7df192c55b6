"""bench.py — TGANv2 training throughput on MI355X (see the driver contract in DESIGN.md §Measurement).

    python bench.py [--gpus N --steps K --warmup W]          (N>1: under torch.distributed.run, or bare — then it starts its
                                                               own N ranks as a child torch.distributed.run before touching a GPU)

Workload (BASELINE.json configs[1]): unconditional TGANv2, 16x64x64x1 clips, per-GPU batch 32, fp32,
RSGAN + zero-centred GP (lambda 0.5), Adam(2e-4, (0.5, 0.999)), 1 D step + 1 G step per iteration,
--subsample_input pyramid 8/16/32/64; synthetic Moving-MNIST-shaped batches resident in HBM.
One "step" = one full iteration of txt2vid_amd.gan.trainer.train_iteration (G fwd, D loss + GP double
backward, Adam, real_pred, G loss backward through D, Adam). Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.6           # 16 x the fp32 matrix rate (dense bf16, no sparsity); only used by --bf16
# reference-executed GFLOP per base sample of one G+D iteration with GP on, keyed by (cond, size, channels)
GFLOP_AS_WRITTEN = {(False, 64, 1): 49.27, (True, 64, 1): 54.9, (False, 128, 3): 196.6, (True, 128, 3): 219.3}


class Params(object):
    frame_sizes = [8, 16, 32, 64]
    subsample_input = True
    discrim_steps = gen_steps = 1
    gp_lambda = 0.5
    no_mean_discrim_loss = no_mean_gen_loss = True


def build_models(dev, seed=100, cond=False, size=64, channels=1):
    if cond:
        from txt2vid_amd.models.tganv2_cond.gen import MultiScaleGen
        from txt2vid_amd.models.tganv2_cond.discrim import MultiScaleDiscrim
    else:
        from txt2vid_amd.models.tganv2.gen import MultiScaleGen
        from txt2vid_amd.models.tganv2.discrim import MultiScaleDiscrim
    from txt2vid_amd.gan.cond_gan import CondGan
    from txt2vid_amd.gan.losses import MixedGanLoss, RSGANLoss
    from txt2vid_amd.optim import Adam
    from txt2vid_amd.util.torch.init import init
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    kw = {'cond_dim': 256} if cond else {}
    gen = MultiScaleGen(width=size, height=size, num_channels=channels, **kw)
    dis = MultiScaleDiscrim(num_channels=channels, **kw)
    init(gen, 'xavier')
    init(dis, 'xavier')
    gen.to(dev).train()
    dis.to(dev).train()
    optD = Adam([{'params': dis.parameters()}], lr=2e-4, betas=(0.5, 0.999))
    optG = Adam([{'params': gen.parameters()}], lr=2e-4, betas=(0.5, 0.999))
    return gen, dis, optD, optG, MixedGanLoss(g_loss=RSGANLoss(), d_loss=RSGANLoss()), CondGan


def synthetic_batches(batch, n, seed, dev, size=64, channels=1):
    from txt2vid_amd.data import SyntheticMovingDigits
    from txt2vid_amd import functional as TF
    ds = SyntheticMovingDigits(length=batch * n, seed=seed, size=size, channels=channels)
    out = []
    for i in range(n):
        # generated in HBM by t2v_synth_clips (bit-identical to the host items ds[i]: tests/test_data_gpu.py)
        vids, _, _ = ds.device_batch(range(i * batch, (i + 1) * batch), dev)         # [B,T,C,H,W]
        out.append(TF.video_to_channel_first(vids))                                   # [B,C,T,H,W]
    ds.check_device_draws()
    return out


def pmc_traffic():
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE are collected in separate `--pmc` runs of tools/conv_micro.py; a PMC pass cannot run inside this
    process). The profile names the commit whose conv.hip it measured; when the kernel source has changed since, the
    figure is dropped (None) rather than reported stale. Returns (bytes, note) or (None, None)."""
    try:
        import glob
        found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
        if not found:
            return None, None
        path = found[-1]                                       # the newest round's passes
        name = 'profiles/' + os.path.basename(path)
        d = json.load(open(path))
        import hashlib
        src = open(os.path.join(ROOT, 'txt2vid_amd', 'csrc', 'conv.hip'), 'rb').read()
        if d.get('conv_hip_sha1') != hashlib.sha1(src).hexdigest():
            return None, '%s was measured on an older conv.hip (sha1 %s): dropped' % (name, d.get('conv_hip_sha1', '?')[:10])
        for k, v in d['kernels'].items():
            if 'conv_igemm_strip3' in k and v.get('FETCH_SIZE_KB_per_launch') and v.get('WRITE_SIZE_KB_per_launch'):
                b = (v['FETCH_SIZE_KB_per_launch'] + v['WRITE_SIZE_KB_per_launch']) * 1024.0
                return b, ('%s: %s, HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 PMC passes; '
                           'average over its forward and masked data-gradient launches); algorithmic %.1f MB; FETCH_SIZE may under-count '
                           'streaming reads by up to 2x on gfx950' % (name, k, d['algorithmic_bytes_per_launch']['strip3<64> fwd / dgrad (x + y + w)'] / 1e6))
    except Exception:
        pass
    return None, None


from txt2vid_amd.util.misc import host_threads          # noqa: E402  (cgroup-aware core count)


def cpu_baseline(threads, budget_s=45.0):
    """The CPU oracle (== reference semantics, pinned by tests/test_oracle_golden.py) on the host cores:
    BASELINE config-1 shape (uncond, B=4, GP on). One warm-up iteration, then timed iterations until the
    time budget is spent (at least 1, at most 3)."""
    from oracle import tganv2_oracle as O
    torch.set_num_threads(threads)
    PG = O.recipe_state(O.gen_shapes(num_channels=1), attn_gamma=0.0)
    PD = O.recipe_state(O.resnet3d_shapes('single_discrim.', 1, 64, 0), attn_gamma=0.0)
    tr = O.OracleTrainer(PG, PD)
    B = 4

    def batch():
        return (torch.rand(B, 16, 1, 64, 64) * 2 - 1).permute(0, 2, 1, 3, 4)
    t0 = time.time()
    tr.step(batch())
    warm = time.time() - t0
    times = []
    while len(times) < 3 and (not times or time.time() - t0 + (sum(times) / len(times)) < budget_s):
        t1 = time.time()
        tr.step(batch())
        times.append(time.time() - t1)
    dt = sum(times) / len(times)
    return {'value': B / dt, 'unit': 'videos/s', 'cores': threads, 'kind': 'port',
            'sample': 'CPU oracle (plain fp32 PyTorch restatement), uncond TGANv2 16x64x64x1, B=4, RSGAN+GP, '
                      '%d timed iteration(s) after 1 warm-up (%.1f s), %.2f s/iter' % (len(times), warm, dt),
            'steps_per_sec': 1.0 / dt}


_LAUNCHER_ENV = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'GROUP_WORLD_SIZE', 'ROLE_RANK', 'ROLE_NAME',
                 'ROLE_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')


def child_env():
    """The environment of a child bench job: this process's, minus what a `torch.distributed.run` parent put there for THIS rank
    (the child is a whole job of its own: `python bench.py --gpus N` starts its own ranks on its own port)."""
    env = {k: v for k, v in os.environ.items() if k not in _LAUNCHER_ENV and not k.startswith('TORCHELASTIC_')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return env


def extra_record(flags, timeout=600, roofline=False):
    """Run `python bench.py <flags>` as a child JOB (no CPU baseline, no roofline passes; with `--gpus N` it starts its own N ranks)
    and return the fields of its JSON line that identify and size the measurement; an error string if it failed or timed out.
    A child job cannot take this process's headline down: a rank that raises or hangs in there only loses the extra record
    (the child's whole process group is killed at the timeout)."""
    import signal
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + flags + ['--no_cpu_baseline', '--no_d_roofline', '--no_extra', '--no_hbm'] + \
        ([] if roofline else ['--no_roofline'])
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=child_env(), start_new_session=True)
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)                  # the exact process group this call started
            p.communicate()
            return {'error': 'child job exceeded %d s and was killed' % timeout}
        line = [l for l in so.decode(errors='replace').splitlines() if l.startswith('{')]
        if p.returncode != 0 or not line:
            return {'error': 'child exited %d: %s' % (p.returncode, se.decode(errors='replace')[-300:])}
        r = json.loads(line[-1])
        keys = ('metric', 'value', 'unit', 'n_gpus', 'ms_per_step', 'steps', 'warmup', 'dtype', 'config', 'final_losses', 'launch_mode',
                'allreduce_ms_per_step', 'allreduce', 'rccl_ranks', 'roofline', 'hbm_bound')
        return {k: r[k] for k in keys if k in r}
    except Exception as e:                                    # never let the extra record take the headline down
        return {'error': repr(e)[:300]}


def sampling_record(gan, batch, dev, iters=20):
    """The metric's second half (eval-mode generated [B,C,16,S,S] videos/s): the calls of `trainer.test`
    (txt2vid/gan/trainer.py:44-90 -> txt2vid_amd/gan/trainer.py:test) minus the PNG writer — eval-mode generator, one latent per
    clip, no sub-sampling, last level only — on the models of the timed run."""
    gan.gen.eval()
    try:
        with torch.no_grad():
            for _ in range(3):
                z = torch.randn(batch, gan.gen.latent_size).to(dev)
                out = gan(z, cond=None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                z = torch.randn(batch, gan.gen.latent_size).to(dev)
                out = gan(z, cond=None)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
        shape = list(out[0].shape)
    finally:
        gan.gen.train()
    return {'videos_per_s': batch / dt, 'ms_per_batch': dt * 1e3, 'batch': batch, 'clip': shape, 'launch_mode': 'eager',
            'gflop_per_video_as_written': 9.95, 'tflops_as_written': 9.95 * batch / dt / 1e3,
            'what': 'eval-mode generator through trainer.test\'s calls (no PNG writer), z drawn on the host per batch as there'}


def cli_record(batch, iters=60, timeout=420):
    """ms per iteration of the REAL command line loop (`python -m txt2vid_amd.train.gan`, the reference's flags: synthetic
    Moving-MNIST clips through the DataLoader + pinned prefetcher + HIP-graph replay, losses read one iteration late), beside
    the bench loop's `ms_per_step`. Child process; parsed from the loop's own `sec/iter` log line (rolling mean of the last 20)."""
    import re
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix='t2v_cli_')
    cfg = os.path.join(tmp, 'synth.json')
    with open(cfg, 'w') as f:
        json.dump({'class': 'txt2vid.data.my_dataset', 'args': {'data': 'synthetic', 'num_frames': 16, 'length': batch * (iters + 4)}}, f)
    cmd = [sys.executable, '-m', 'txt2vid_amd.train.gan', '--data', cfg, '--num_channels', '1', '--cuda', '--frame_sizes', '8', '16', '32', '64',
           '--D_names', 'video', '--G_lr', '0.0002', '--D_lr', '0.0002', '--D_beta1', '0.5', '--D_beta2', '.999', '--G_beta1', '0.5',
           '--G_beta2', '.999', '--D_loss', 'txt2vid.gan.losses.RSGANLoss', '--init_method', 'xavier', '--discrim_steps', '1', '--seed', '100',
           '--gp_lambda', '.5', '--subsample_input', '--workers', '4', '--log_period', '20', '--G', 'txt2vid.models.tganv2.gen.MultiScaleGen',
           '--D', 'txt2vid.models.tganv2.discrim.MultiScaleDiscrim', '--dont_use_sent', '--batch_size', str(batch), '--epochs', '1',
           '--out', os.path.join(tmp, 'out'), '--out_samples', os.path.join(tmp, 'samples'), '--max_iters', str(iters),
           '--save_model_period', '0', '--save_example_period', '0']
    try:
        p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, env=child_env())
        out = p.stdout.decode(errors='replace')
        m = re.findall(r'Iter (\d+),.*? - ([0-9.]+) sec/iter; ([0-9.]+) sec/batch load', out)
        if p.returncode != 0 or not m:
            return {'error': 'CLI exited %d: %s' % (p.returncode, out[-300:])}
        it, sec, load = m[-1]
        return {'ms_per_iter': float(sec) * 1e3, 'wait_for_batch_ms': float(load) * 1e3, 'at_iteration': int(it), 'batch': batch,
                'what': 'python -m txt2vid_amd.train.gan (reference flags, synthetic clips, 4 loader workers, graph replay), mean of the last 20 iterations'}
    except Exception as e:
        return {'error': repr(e)[:300]}
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch(n):
    """One process per GPU over RCCL, as the driver's own N>1 command line does it:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same flags>."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL's intra-node transport needs it on this host driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write('[bench] launching %d ranks: %s\n' % (n, ' '.join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def per_tile_breakdown(path, steps, peak, kind):
    """Forward / data-gradient GEMM launches of the instrumented iterations grouped by kernel family: launches and GPU ms per
    iteration, achieved TFLOP/s on the executed FLOPs, fraction of the MFMA peak (from the library's per-launch records)."""
    import csv
    names = {0: 'igemm', 1: 'strip', 5: 'strip3', 9: 'pool fwd', 10: 'pool dgrad'}
    agg = {}
    try:
        with open(path) as f:
            for r in csv.DictReader(f):
                if int(r['kind']) != kind:
                    continue
                plan = [int(v) for v in r['plan'].split(':')]
                if plan[0] not in names:
                    continue
                key = '%s %dx%d' % (names[plan[0]], plan[1], plan[2])
                a = agg.setdefault(key, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += float(r['ms'])
                a[2] += float(r['flops'])
    except (OSError, KeyError, ValueError):
        return None
    out = {}
    for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        out[key] = {'launches_per_step': n / steps, 'gpu_ms_per_step': ms / steps, 'achieved': tf, 'frac': tf / peak}
    return out


class _Run(object):
    pass


def timed_run(args, cond, size, channels, steps, warmup, rank, world, dev, log, prof_eager=False):
    """Build the models of one workload (current conv precision), warm up (2 eager iterations + graph capture), then time exactly
    `steps` iterations between barrier + synchronize pairs. Returns the objects the roofline passes need plus `dt` (this rank's
    wall time of the timed region), the final losses and — for world > 1 — the event-timed gradient exchanges."""
    from txt2vid_amd import dist as tdist
    from txt2vid_amd import functional as TF
    from txt2vid_amd._lib import lib
    import torch.distributed as dist
    from txt2vid_amd.gan.trainer import train_iteration, GraphedTrainStep
    R = _Run()
    gen, dis, optD, optG, losses, CondGan = build_models(dev, cond=cond, size=size, channels=channels)
    txt = tokens = None
    if cond:                    # Bi-LSTM sentence encoder (random init) + 8-token synthetic captions
        from txt2vid_amd.data import Vocab
        from txt2vid_amd.models.txt.basic import Seq2Seq
        from txt2vid_amd.util.torch.init import init
        txt = Seq2Seq(vocab_size=len(Vocab()))
        init(txt, 'xavier')
        txt.to(dev)
        tokens = torch.randint(4, len(Vocab()), (args.batch, 8)).to(dev)
    gan = CondGan(gen=gen, discrims=[dis], cond_encoder=txt, discrim_names=['video'], gp_scale=float(world))
    grad_sync = None
    if world > 1:
        arenas = {'D': tdist.model_arena(dis, TF.copy_into), 'G': tdist.model_arena(gen, TF.copy_into)}
        grad_sync = tdist.make_grad_sync(arenas, {'D': optD, 'G': optG}, world)
    prm = Params()
    prm.frame_sizes = [size // 8, size // 4, size // 2, size]
    pool = synthetic_batches(args.batch, 4, 100 + rank, dev, size, channels)
    random.seed(100 + rank)
    np.random.seed(100 + rank)
    torch.manual_seed(100 + rank)

    graphed = None
    if not args.eager:
        graphed = GraphedTrainStep(gan, optD, optG, losses, prm, dev, tuple(pool[0].shape), grad_sync=grad_sync, warmup=2,
                                   cond_dim=256 if cond else 0)
    graphed_txt = None
    if txt is not None and not args.eager:
        from txt2vid_amd.gan.trainer import GraphedSentenceEncoder
        graphed_txt = GraphedSentenceEncoder(txt, dev)

    def sentence_codes():            # part of every iteration: its own HIP graph per caption length (lengths vary in real data)
        if txt is None:
            return None
        if graphed_txt is not None:
            return graphed_txt.encode(tokens, [8] * args.batch)
        return txt.encode(tokens, [8] * args.batch)[2].detach()

    def step(i):
        if graphed is not None:
            return graphed.step(pool[i % len(pool)], sentence_codes()) + (None, None)
        return train_iteration(gan, pool[i % len(pool)], sentence_codes(), optD, optG, losses, prm, dev, grad_sync=grad_sync)

    if graphed is not None and warmup < 3:
        warmup = 3                           # 2 eager iterations + the capture must precede the timed region
    log('models + data ready; warm-up')
    for i in range(warmup):
        step(i)
        torch.cuda.synchronize()
        log('warm-up step %d done' % i)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    R.prof_in_region = prof_eager and graphed is None        # eager launches: events bracket them inside the timed region
    barrier()
    if R.prof_in_region:
        lib().t2v_prof_begin(min(1 << 16, 4096 * steps))
        log('instrumentation ready')
    if grad_sync is not None:
        grad_sync.time_exchanges(True)
    t0 = time.perf_counter()
    for i in range(steps):
        lD, lG, _, _ = step(warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    R.dt = time.perf_counter() - t0
    R.exchange = None
    if grad_sync is not None:
        R.exchange = grad_sync.exchange_ms()
        grad_sync.time_exchanges(False)
    R.lD, R.lG = float(lD), float(lG)
    R.gan, R.optD, R.optG, R.losses, R.prm, R.pool, R.graphed, R.grad_sync, R.txt = gan, optD, optG, losses, prm, pool, graphed, grad_sync, txt
    R.sentence_codes, R.warmup = sentence_codes, warmup
    return R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (weak scaling)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_roofline', action='store_true')
    ap.add_argument('--no_d_roofline', action='store_true', help='skip the D forward+backward roofline pass')
    ap.add_argument('--eager', action='store_true', help='no HIP-graph replay (eager launches)')
    ap.add_argument('--bf16', action='store_true', help='bf16-compute mode (forward / data-gradient GEMMs on bf16 MFMA, fp32 storage and '
                                                       'accumulation): an extra, NOT the precision the metric is quoted on')
    ap.add_argument('--cond', action='store_true', help='text-conditioned TGANv2 (BASELINE configs[2] shape) instead of '
                                                       'the unconditional configs[1] workload the metric is quoted on; with --bf16 = configs[2]')
    ap.add_argument('--size', type=int, default=64, choices=(64, 128), help='frame side; 128 with --channels 3 --cond --bf16 --batch 16 '
                                                                          '= the per-GPU share of BASELINE configs[4] (MSRVDC shape)')
    ap.add_argument('--channels', type=int, default=1, choices=(1, 3))
    ap.add_argument('--no_hbm', action='store_true', help='skip the HBM-bound kernels block')
    ap.add_argument('--no_extra', action='store_true', help='skip the extra BASELINE configs[2] record (text-conditioned, bf16 compute) of the default run')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` (no launcher): start the N ranks as a CHILD torch.distributed.run — nothing in this
        # process has touched the GPU yet (importing torch does not), and it never does: it only relays the ranks' output
        # (rank 0 prints the JSON line) and exits with their return code.
        raise SystemExit(self_launch(args.gpus))

    from txt2vid_amd import dist as tdist
    from txt2vid_amd import functional as TF
    from txt2vid_amd._lib import lib
    import torch.distributed as dist

    if args.bf16:
        TF.set_conv_precision('bf16')
    bf16 = TF.CONV_PRECISION == 'bf16'
    rank, world = tdist.init_from_env('nccl')
    from txt2vid_amd.util.misc import limit_host_threads
    limit_host_threads()                 # the default CPU pool (one spinning thread per logical CPU) starves the HIP runtime
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    local = tdist.local_device_index()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    T0 = time.perf_counter()

    def log(msg):
        if rank == 0:
            sys.stderr.write('[bench %.1fs] %s\n' % (time.perf_counter() - T0, msg))
            sys.stderr.flush()

    prof = not args.no_roofline
    R = timed_run(args, args.cond, args.size, args.channels, args.steps, args.warmup, rank, world, dev, log, prof_eager=prof)
    args.warmup = R.warmup
    gan, optD, optG, losses, prm, pool, graphed, grad_sync, txt = R.gan, R.optD, R.optG, R.losses, R.prm, R.pool, R.graphed, R.grad_sync, R.txt
    sentence_codes, dt, lD, lG, prof_in_region = R.sentence_codes, R.dt, R.lD, R.lG, R.prof_in_region
    from txt2vid_amd.gan.trainer import train_iteration
    log('timed region done: %.1f ms/step' % (dt / args.steps * 1e3))
    lD, lG = float(lD), float(lG)
    roof = None
    prof_steps, prof_dt = args.steps, dt
    if prof and not prof_in_region:
        # A replayed HIP graph re-issues no launch calls, so per-launch hipEvents cannot be recorded inside
        # it: the same iterations are re-run eagerly right after the timed region with the events on.
        prof_steps = min(args.steps, 5)
        torch.cuda.synchronize()
        lib().t2v_prof_begin(min(1 << 16, 4096 * prof_steps))
        t1 = time.perf_counter()
        import contextlib
        # (on the stream the graphs were captured on: the parameters' AccumulateGrad nodes remember it)
        side = torch.cuda.stream(graphed.side) if graphed is not None else contextlib.nullcontext()
        if graphed is not None:
            graphed.release()            # the replay is over: the gradient sink may recycle its table slots in the eager pass
            graphed.side.wait_stream(torch.cuda.current_stream())
        with side:
            for i in range(prof_steps):
                train_iteration(gan, pool[i % len(pool)], sentence_codes(), optD, optG, losses, prm, dev, grad_sync=grad_sync)
        torch.cuda.synchronize()
        prof_dt = time.perf_counter() - t1
        log('instrumented eager pass done: %.1f ms/step' % (prof_dt / prof_steps * 1e3))
    if prof:
        out = (C.c_double * 18)()
        dump = os.environ.get('T2V_PROF_DUMP')
        own_dump = dump is None
        if own_dump:                               # per-launch records (kernel plan, time, FLOPs) for the per-tile breakdown below
            import tempfile
            dump = os.path.join(tempfile.gettempdir(), 't2v_prof_%d.csv' % os.getpid())
            os.environ['T2V_PROF_DUMP'] = dump
        over = lib().t2v_prof_end(out, 6)          # kinds: 0 igemm, 1 wgrad, 2 wgrad reduce, 3 thin convs, 4 split-K reduce, 5 bf16 igemm
        by_tile = per_tile_breakdown(dump, prof_steps, PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS, 5 if bf16 else 0)
        os.environ.pop('T2V_PROF_DUMP', None)      # (the later roofline passes must not overwrite a dump the caller asked for)
        if own_dump:
            try:
                os.remove(dump)
            except OSError:
                pass
        conv_flops = out[1] + out[4] + out[10] + out[16]      # fp32 implicit GEMM + weight gradients + thin kernels + bf16 GEMM
        if bf16:                                   # bf16-compute mode: the dominant kernel is the bf16 GEMM, priced against the bf16 peak
            out[0], out[1], out[2] = out[15], out[16], out[17]
        ms, fl, cnt = out[0] + out[12], out[1], out[2]
        if cnt > 0 and ms > 0:
            ach = fl / (ms * 1e-3) / 1e12
            peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
            roof = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': ach / peak, 'traffic': pmc_traffic()[0], 'traffic_note': pmc_traffic()[1],
                    'kernel': ('conv_igemm_bf16_kernel (implicit-GEMM conv forward + data-gradient, bf16 MFMA 32x32x16, fp32 tensors; '
                               'includes its split-K reduce pass)' if bf16 else
                               'conv_igemm_kernel (implicit-GEMM conv forward + data-gradient, fp32 MFMA 32x32x2; '
                               'includes its split-K reduce pass)'),
                    'launches_per_step': cnt / prof_steps, 'avg_launch_us': out[0] * 1e3 / cnt,
                    'splitk_reduce': {'launches_per_step': out[14] / prof_steps, 'gpu_ms_per_step': out[12] / prof_steps},
                    'timing': 'hipExtLaunchKernelGGL start/stop events = the dispatch\'s own begin/end (same clock as the rocprofv3 kernel trace)',
                    'flops_counted': 'executed MACs x2 (padding-only taps excluded), summed over all launches',
                    'measured_over': ('the timed region (eager launches)' if prof_in_region else
                                      '%d eager iterations right after the timed region (graph replay has no launch calls to '
                                      'bracket)' % prof_steps),
                    'gpu_ms_per_step': ms / prof_steps, 'pool_overflow': bool(over),
                    # every convolution FLOP the iteration executes (forward / data / weight gradients, thin kernels) over the
                    # WHOLE step time: what the step as a whole achieves against the matrix peak (kernel fraction above: the
                    # implicit-GEMM launches alone over their own time)
                    'all_in_frac': conv_flops / prof_steps / (dt / args.steps) / 1e12 / peak,
                    # the same launches by kernel family (tile): `frac` above is their time-weighted aggregate
                    'by_tile': by_tile,
                    # the single kernel family with the most GPU time among them (what `traffic` was measured on)
                    'dominant_kernel': (dict(max(by_tile.items(), key=lambda kv: kv[1]['gpu_ms_per_step'])[1],
                                             family=max(by_tile.items(), key=lambda kv: kv[1]['gpu_ms_per_step'])[0]) if by_tile else None),
                    'executed_conv_tflop_per_step': conv_flops / prof_steps / 1e12,
                    'wgrad': {'achieved': (out[4] / (out[3] * 1e-3) / 1e12) if out[3] > 0 else None,
                              'launches_per_step': out[5] / prof_steps, 'gpu_ms_per_step': out[3] / prof_steps,
                              'reduce_gpu_ms_per_step': out[6] / prof_steps}}
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    gb = args.batch * world
    default_shape = args.size == 64 and args.channels == 1
    if not default_shape:
        workload = ('%s TGANv2 at the MSRVDC shape of BASELINE configs[4] ' % ('text-conditioned' if args.cond else 'unconditional'))
    elif args.cond:
        workload = ('BASELINE configs[2]%s: text-conditioned TGANv2 (Bi-LSTM sentence codes, 2-D + 3-D non-local blocks) ' %
                    ('' if bf16 else ' shape at fp32'))
    else:
        workload = 'BASELINE configs[1]: unconditional TGANv2 '
    workload += ('16x%dx%dx%d, per-GPU batch %d, %s, RSGAN + GP 0.5, Adam 2e-4 (0.5,0.999), 1 D + 1 G step, subsample_input pyramid %s'
                 % (args.size, args.size, args.channels, args.batch, 'bf16 compute' if bf16 else 'fp32', '/'.join(map(str, prm.frame_sizes))))
    gf = GFLOP_AS_WRITTEN.get((bool(args.cond), args.size, args.channels))
    res = {
        'metric': 'TGANv2 GAN train throughput (G+D steps x global batch), 16x%dx%d videos/sec' % (args.size, args.size),
        'value': gb * args.steps / dt, 'unit': 'videos/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'steps_per_sec': args.steps / dt, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16 compute (forward / data-gradient / 3-tap weight-gradient MFMA), f32 storage + accumulation' if bf16 else 'f32',
        'data': 'synthetic',
        'config': {'workload': workload,
                   'global_batch': gb, 'per_gpu_batch': args.batch, 'parallelism': 'dp%d' % world,
                   'as_written_tflop_per_step': gf * gb / 1e3 if gf else None},
        'final_losses': {'lossD': lD, 'lossG': lG}, 'launch_mode': 'eager' if graphed is None else 'hip-graph replay (%d graphs/step)' % (graphed.n_graphs + (1 if txt is not None else 0)),
        'as_written_tflops': gf * gb * args.steps / dt / 1e3 if gf else None,
    }
    if grad_sync is not None:
        res['config']['grad_exchange_mb_per_step'] = sum(a.exchanged_bytes() for a in grad_sync.arenas.values()) / 1e6
        # the two per-step collectives (D gradients, G gradients), event-timed on the issuing stream inside the timed region,
        # max over ranks; `rccl_ranks` = ranks of the process group when its backend is nccl (= RCCL on ROCm), else 0
        ex = R.exchange or {'D': 0.0, 'G': 0.0, 'n': 0}
        tx = torch.tensor([ex['D'], ex['G']], device=dev, dtype=torch.float64)
        dist.all_reduce(tx, op=dist.ReduceOp.MAX)
        res['allreduce_ms_per_step'] = float(tx.sum().item()) / args.steps
        res['allreduce'] = {'D_ms_per_step': float(tx[0].item()) / args.steps, 'G_ms_per_step': float(tx[1].item()) / args.steps,
                            'collectives_per_step': ex['n'] / args.steps, 'backend': dist.get_backend(),
                            'overlap': 'none: D\'s Adam needs every reduced D gradient before the next kernel, G\'s likewise (DESIGN §6)'}
        res['rccl_ranks'] = dist.get_world_size() if dist.get_backend() == 'nccl' else 0
    if roof is not None:
        if gf:
            # the REFERENCE-executed FLOP of the iteration (what the reference's kernels compute, incl. the ConvLSTM's dead
            # taps and the stem conv2's odd frames that this implementation does not execute) over the whole step time: the figure
            # that does not move when work is removed rather than sped up
            roof['as_written_all_in_frac'] = gf * 1e9 * (gb / world) / (dt / args.steps) / 1e12 / roof['peak']
        res['roofline'] = roof
    if rank == 0 and world == 1 and prof and not args.no_d_roofline and default_shape:
        # the north-star's own target line: D forward+backward on un-subsampled 16x64x64 clips
        from txt2vid_amd.util.roofline import d_fwdbwd_roofline
        log('D forward+backward roofline pass')
        try:
            res['d_fwdbwd_roofline'] = d_fwdbwd_roofline(batch=args.batch, iters=3, device=dev)
        except Exception as e:                                    # side measurements never take the headline down
            res['d_fwdbwd_roofline'] = {'error': repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_extra and not args.cond and not bf16 and default_shape:
        log('sampling (eval-mode generator) and the CLI loop')
        try:
            res['sampling'] = sampling_record(gan, args.batch, dev)
        except Exception as e:                                    # side measurements never take the headline down
            res['sampling'] = {'error': repr(e)[:300]}
        if not args.eager:
            try:
                res['cli_ms_per_iter'] = cli_record(args.batch)
            except Exception as e:
                res['cli_ms_per_iter'] = {'error': repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.cond and default_shape:
        log('timing the CPU oracle on %d host threads' % host_threads())
        try:
            res['cpu_baseline'] = cpu_baseline(host_threads())
        except Exception as e:
            res['cpu_baseline'] = {'error': repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_extra and not args.cond and not bf16 and default_shape and not args.eager:
        # BASELINE configs[2] (text-conditioned, Bi-LSTM sentence codes, non-local blocks on, bf16 compute) as an EXTRA record of the
        # same line: measured by a child process after this one's timed region, never part of `value`
        log('extra record: BASELINE configs[2] (child process)')
        # (with its own `roofline` object: the bf16 GEMM launches over their own time against the dense bf16 matrix peak the line states)
        res['extra_records'] = {'configs[2]': extra_record(['--cond', '--bf16', '--batch', str(args.batch), '--steps', '10', '--warmup', '3'],
                                                           roofline=True)}
    if rank == 0 and world == 1 and prof and not args.no_hbm and default_shape and not args.cond and not bf16:
        # the HBM-bound sub-operations, each in isolation at the benchmark's shapes: GB/s on algorithmic bytes vs ~8 TB/s
        from txt2vid_amd.util.roofline import hbm_bound_lines
        log('HBM-bound kernels in isolation')
        try:
            res['hbm_bound'] = hbm_bound_lines(device=dev, batch=args.batch)
            import glob
            pmc = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm.json')))
            if pmc:
                res['hbm_bound']['pmc'] = 'profiles/%s (rocprofv3 FETCH_SIZE / WRITE_SIZE per launch of the same micro-benchmark)' % os.path.basename(pmc[-1])
        except Exception as e:                                    # never let a side measurement take the headline down
            res['hbm_bound'] = {'error': repr(e)[:300]}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()     # the headline is complete: nothing below takes part in a collective of THIS job
    if world > 1 and rank == 0 and not args.no_extra and not args.cond and not bf16 and default_shape and not args.eager:
        # BASELINE configs[3] (the scaling config: text-conditioned, bf16 compute, per-GPU batch 32, RCCL data parallel) as an EXTRA
        # record beside the configs[1] value — never part of `value`. It runs as a CHILD JOB with its own ranks (the other ranks of
        # this job have left by now): a rank of it that raises or hangs costs the extra record, never the headline line below.
        log('extra record: BASELINE configs[3] as a child job on %d ranks' % world)
        del R, gan, optD, optG, graphed, grad_sync, pool
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        res.setdefault('extra_records', {})['configs[3]'] = extra_record(
            ['--gpus', str(world), '--cond', '--bf16', '--batch', str(args.batch), '--steps', '10', '--warmup', '3'], timeout=420)
    if rank == 0:
        print(json.dumps(res))


if __name__ == '__main__':
    main()
