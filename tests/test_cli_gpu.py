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
    assert glob.glob(os.path.join(smp2, 'fake_0_0.png'))


def test_cond_cli_train(tmp_path):
    out_dir, smp = str(tmp_path / 'out'), str(tmp_path / 'samples')
    log = run(COMMON + ['--G', 'txt2vid.models.tganv2_cond.gen.MultiScaleGen', '--D', 'txt2vid.models.tganv2_cond.discrim.MultiScaleDiscrim',
                        '--sent', 'txt2vid.models.txt.basic.Seq2Seq', '--batch_size', '8', '--epochs', '1', '--out', out_dir,
                        '--out_samples', smp, '--max_iters', '2', '--save_model_period', '1000', '--save_example_period', '2'], tmp_path)
    assert 'Iter 2, Loss_D' in log
    assert glob.glob(os.path.join(smp, 'sentences_epoch000_iter_000002.txt'))
