"""GPU: the reference's command line drives this build end to end (train a few iterations, checkpoint,
reload, sample), through `python -m txt2vid_amd.train.gan` with the reference's class names and flags."""
import glob
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, tmp):
    cfg = os.path.join(str(tmp), 'synth.json')          # same shape as the reference's config/synth.json
    with open(cfg, 'w') as f:
        json.dump({'class': 'txt2vid.data.my_dataset', 'args': {'data': 'synthetic', 'num_frames': 16, 'length': 64}}, f)
    cmd = [sys.executable, '-m', 'txt2vid_amd.train.gan', '--data', cfg] + args
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode(errors='replace')
    assert p.returncode == 0, out[-3000:]
    return out


COMMON = ['--num_channels', '1', '--cuda', '--frame_sizes', '8', '16', '32', '64', '--D_names', 'video',
          '--G_lr', '0.0002', '--D_lr', '0.0002', '--D_beta1', '0.5', '--D_beta2', '.999', '--G_beta1', '0.5', '--G_beta2', '.999',
          '--D_loss', 'txt2vid.gan.losses.RSGANLoss', '--init_method', 'xavier', '--discrim_steps', '1', '--seed', '100',
          '--gp_lambda', '.5', '--subsample_input', '--workers', '0', '--log_period', '1']


def test_uncond_cli_train_checkpoint_resume_and_sample(tmp_path):
    out_dir, smp = str(tmp_path / 'out'), str(tmp_path / 'samples')
    base = COMMON + ['--G', 'txt2vid.models.tganv2.gen.MultiScaleGen', '--D', 'txt2vid.models.tganv2.discrim.MultiScaleDiscrim',
                     '--dont_use_sent', '--batch_size', '8', '--epochs', '1', '--out', out_dir, '--out_samples', smp]
    log = run(base + ['--max_iters', '3', '--save_model_period', '3', '--save_example_period', '3'], tmp_path)
    assert 'Iter 3, Loss_D' in log
    ck = glob.glob(os.path.join(out_dir, 'iter_3_lossG_*'))
    assert len(ck) == 1
    pngs = glob.glob(os.path.join(smp, 'fake_samples_epoch_000_iter_000003_*.png'))
    assert len(pngs) == 4 and os.path.exists(os.path.join(smp, 'real_samples.png'))
    with open(pngs[0], 'rb') as f:
        assert f.read(8) == b'\x89PNG\r\n\x1a\n'
    # resume from the checkpoint (state_dict layout = the reference's) and keep training
    log2 = run(base + ['--max_iters', '2', '--weights', ck[0], '--save_model_period', '1000', '--save_example_period', '0'], tmp_path)
    assert 'Iter 2, Loss_D' in log2
    # sampling path (trainer.test): eval-mode generator renders full 16x64x64 clips
    smp2 = str(tmp_path / 'samples_test')
    run(base[:-2] + ['--out_samples', smp2, '--test', '--weights', ck[0], '--num_samples', '1', '--max_iters', '1'], tmp_path)
    assert os.path.exists(os.path.join(smp2, 'real_0.png')) and os.path.exists(os.path.join(smp2, '64x64_0_0.jpg'))      # trainer.py:76,88
    assert len(os.listdir(smp2)) == 2                                     # one batch per sample, last level only in eval mode


def test_cond_cli_train(tmp_path):
    out_dir, smp = str(tmp_path / 'out'), str(tmp_path / 'samples')
    log = run(COMMON + ['--G', 'txt2vid.models.tganv2_cond.gen.MultiScaleGen', '--D', 'txt2vid.models.tganv2_cond.discrim.MultiScaleDiscrim',
                        '--sent', 'txt2vid.models.txt.basic.Seq2Seq', '--batch_size', '8', '--epochs', '1', '--out', out_dir,
                        '--out_samples', smp, '--max_iters', '2', '--save_model_period', '1000', '--save_example_period', '2'], tmp_path)
    assert 'Iter 2, Loss_D' in log
    assert glob.glob(os.path.join(smp, 'sentences_epoch000_iter_000002.txt'))


def test_pretrained_sentence_encoder_feeds_the_graph_replayed_gan_loop(tmp_path):
    """a checkpoint written by `train/txt.py` (the whole Seq2Seq pickled, reference layout) loaded through
    `train/gan.py --sent_weights` and used under HIP-graph replay — the encoder must run on the forward-only, capturable
    kernels whatever mode it was pickled in."""
    import pickle
    import random
    import torch
    from txt2vid_amd.data import Vocab
    vocab = Vocab()
    words = [w for w in vocab.word2idx if not w.startswith('<')] if hasattr(vocab, 'word2idx') else ['digit', '0', 'is', 'left']
    rng = random.Random(3)
    sents = {'v%d' % i: [' '.join(rng.choice(words) for _ in range(rng.randint(3, 6)))] for i in range(24)}
    with open(tmp_path / 'sents.pkl', 'wb') as f:
        pickle.dump(sents, f)
    with open(tmp_path / 'vocab.pkl', 'wb') as f:
        pickle.dump(vocab, f)
    out_txt = tmp_path / 'txt_out'
    cmd = [sys.executable, '-m', 'txt2vid_amd.train.txt', '--data', str(tmp_path / 'sents.pkl'), '--vocab', str(tmp_path / 'vocab.pkl'),
           '--out', str(out_txt), '--cuda', '--seed', '5', '--batch_size', '8', '--epoch', '4', '--workers', '0', '--max_iters', '4',
           '--save_model_period', '4', '--log_period', '2']
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors='replace')[-3000:]
    ck = sorted(glob.glob(os.path.join(str(out_txt), 'iter_4_*')))
    assert len(ck) == 1
    saved = torch.load(ck[0], weights_only=False, map_location='cpu')
    assert saved['txt'].encoder.with_grad is False                   # pickled in forward-only mode
    # ... and a pickle that DOES carry the autograd-path flag (what round 1 wrote) must still work
    saved['txt'].differentiable(True)
    legacy = str(tmp_path / 'legacy_ckpt')
    torch.save(saved, legacy)
    out_dir, smp = str(tmp_path / 'out'), str(tmp_path / 'samples')
    for weights in (ck[0], legacy):
        log = run(COMMON + ['--G', 'txt2vid.models.tganv2_cond.gen.MultiScaleGen', '--D', 'txt2vid.models.tganv2_cond.discrim.MultiScaleDiscrim',
                            '--sent_weights', weights, '--vocab', str(tmp_path / 'vocab.pkl'), '--batch_size', '8', '--epochs', '1',
                            '--out', out_dir, '--out_samples', smp, '--max_iters', '5', '--save_model_period', '1000',
                            '--save_example_period', '0'], tmp_path)
        assert 'Iter 5, Loss_D' in log and 'HIP-graph replay' in log


def test_cli_resumes_from_checkpoints_written_by_the_real_reference(tmp_path):
    """`--weights` on tests/golden/ref_checkpoint_uncond.pt and `--weights` + `--sent_weights` on the text-conditioned pair
    (ref_checkpoint_cond.pt, ref_seq2seq.pt): files produced by the reference's own CondGan.save_dict / Adam.state_dict /
    whole-module Seq2Seq pickle (make_golden.py; one element per tensor kept), resumed through this build's CLI with the
    iteration under HIP-graph replay (closes SURVEY §8 f3 "load reference-produced checkpoints")."""
    import re
    gold = os.path.join(ROOT, 'tests', 'golden')
    out_dir, smp = str(tmp_path / 'out'), str(tmp_path / 'samples')
    tail = ['--batch_size', '8', '--epochs', '1', '--out', out_dir, '--out_samples', smp, '--max_iters', '4',
            '--save_model_period', '1000', '--save_example_period', '0']
    log = run(COMMON + ['--G', 'txt2vid.models.tganv2.gen.MultiScaleGen', '--D', 'txt2vid.models.tganv2.discrim.MultiScaleDiscrim',
                        '--dont_use_sent', '--weights', os.path.join(gold, 'ref_checkpoint_uncond.pt')] + tail, tmp_path)
    m = re.search(r'Iter 4, Loss_D: ([-0-9.naninf]+) Loss_G: ([-0-9.naninf]+)', log)
    assert m and all(abs(float(v)) < 1e3 for v in m.groups()), log[-1500:]
    log = run(COMMON + ['--G', 'txt2vid.models.tganv2_cond.gen.MultiScaleGen', '--D', 'txt2vid.models.tganv2_cond.discrim.MultiScaleDiscrim',
                        '--sent_weights', os.path.join(gold, 'ref_seq2seq.pt'),
                        '--weights', os.path.join(gold, 'ref_checkpoint_cond.pt')] + tail, tmp_path)
    m = re.search(r'Iter 4, Loss_D: ([-0-9.naninf]+) Loss_G: ([-0-9.naninf]+)', log)
    assert m and all(abs(float(v)) < 1e3 for v in m.groups()), log[-1500:]
    assert 'HIP-graph replay' in log


def test_cond_cli_end2end(tmp_path):
    """`--end2end` through the CLI (train/gan.py:82-85): the text encoder in both optimisers, eager launches (the encoder graph spans
    the D and the G step), three iterations without a version-counter error and with finite losses."""
    import re
    out_dir, smp = str(tmp_path / 'out'), str(tmp_path / 'samples')
    log = run(COMMON + ['--G', 'txt2vid.models.tganv2_cond.gen.MultiScaleGen', '--D', 'txt2vid.models.tganv2_cond.discrim.MultiScaleDiscrim',
                        '--sent', 'txt2vid.models.txt.basic.Seq2Seq', '--end2end', '--batch_size', '8', '--epochs', '1', '--out', out_dir,
                        '--out_samples', smp, '--max_iters', '3', '--save_model_period', '1000', '--save_example_period', '0'], tmp_path)
    m = re.search(r'Iter 3, Loss_D: ([-0-9.naninf]+) Loss_G: ([-0-9.naninf]+)', log)
    assert m and all(abs(float(v)) < 1e3 for v in m.groups()), log[-1500:]
    assert 'HIP-graph replay' not in log
