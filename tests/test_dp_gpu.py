"""GPU: the data-parallel path end to end on one card — two ranks (gloo, both on device 0) run the graph-replayed
iteration with the gradient sink and the per-step arena all-reduce (tools/dp_check.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_replicas_stay_identical():
    env = dict(os.environ, T2V_DIST_BACKEND='gloo', T2V_SINGLE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29531', os.path.join(ROOT, 'tools', 'dp_check.py')]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode(errors='replace')
    assert p.returncode == 0, out[-3000:]
    line = [l for l in out.splitlines() if l.startswith('DP_CHECK')]
    assert line and 'identical after 5 iterations: True' in line[0], out[-2000:]


def test_two_rank_nccl_replicas_stay_identical():
    """The same check over RCCL, one rank per GPU — only where the box has two GPUs (the builder's lease has one; the
    driver's multi-GPU node runs it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (RCCL refuses two ranks on one device)')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('T2V_DIST_BACKEND', None)
    env.pop('T2V_SINGLE_DEVICE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'tools', 'dp_check.py')]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors='replace')
    assert p.returncode == 0, out[-3000:]
    line = [l for l in out.splitlines() if l.startswith('DP_CHECK')]
    assert line and 'identical after 5 iterations: True' in line[0], out[-2000:]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls the N=1 case): bench.py starts the two
    ranks itself as a child torch.distributed.run before touching the GPU, relays rank 0's JSON line and exit code.
    Rehearsed here with gloo and both ranks on device 0 (one GPU per lease); over RCCL when the box has two GPUs."""
    import json
    import numpy as np
    import torch
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    if torch.cuda.device_count() < 2:
        env.update(T2V_DIST_BACKEND='gloo', T2V_SINGLE_DEVICE='1')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '3', '--batch', '8',
           '--no_cpu_baseline', '--no_roofline']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors='replace')[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 16 and res['scaling'] == 'weak'
    assert res['config']['grad_exchange_mb_per_step'] > 100          # D 116 MB + G's live 76 MB
    # the two per-step collectives are event-timed inside the timed region; `rccl_ranks` says whether they ran over RCCL
    assert res['allreduce_ms_per_step'] > 0 and res['allreduce']['collectives_per_step'] >= 2
    assert res['rccl_ranks'] == (2 if torch.cuda.device_count() >= 2 else 0)
    # BASELINE's scaling config (configs[3]: text-conditioned, bf16 compute) rides along as an extra record on the same ranks
    c3 = res['extra_records']['configs[3]']
    assert 'error' not in c3, c3
    assert c3['n_gpus'] == 2 and c3['value'] > 0 and c3['allreduce_ms_per_step'] > 0 and np.isfinite(c3['final_losses']['lossD'])
    assert res['value'] > 0 and res['steps'] == 2


def test_arena_gather_copies_dense_gradients_into_tap_major_slots():
    """`GradArena.gather()` on the GPU copy path. A tap-major master weight (ConvLSTM) whose gradient did NOT come
    through the sink arrives dense [Cout,Cin,kh,kw]; it must land in the arena's [kh][kw][Cout][Cin] memory order (the flat copy
    kernel alone would scramble it), and the live-tap rows the exchange packs must be the right taps."""
    import torch
    from txt2vid_amd import functional as TF
    from txt2vid_amd.dist import GradArena
    dev = 'cuda:0'
    w = torch.nn.Parameter(TF.tap_major(torch.zeros(8, 6, 3, 3, device=dev)))
    b = torch.nn.Parameter(torch.zeros(5, device=dev))
    d = torch.nn.Parameter(torch.zeros(3, 2, 3, 3, device=dev))                 # ordinary dense weight
    arena = GradArena([b, w, d], TF.copy_into, live_taps={w: [4]})
    g = torch.arange(8 * 6 * 9, dtype=torch.float32, device=dev).view(8, 6, 3, 3)
    w.grad, b.grad, d.grad = g.clone(), torch.ones(5, device=dev), torch.full((3, 2, 3, 3), 2.0, device=dev)
    arena.gather()
    torch.cuda.synchronize()
    views = dict(zip([id(p) for p in arena.params], arena.views()))
    assert torch.equal(views[id(w)], g) and views[id(w)].stride() == w.stride()
    assert torch.equal(views[id(d)], d.grad) and torch.equal(views[id(b)], b.grad)
    arena._taps(True)                                                           # pack the live tap as the exchange does
    torch.cuda.synchronize()
    assert torch.equal(arena.compact.view(8, 6), g[:, :, 1, 1])
    # a tap-major gradient (same strides, other memory) takes the stride-aware path too
    w.grad = TF.tap_major(g * 3)
    arena.gather()
    torch.cuda.synchronize()
    assert torch.equal(arena.views()[[i for i, q in enumerate(arena.params) if q is w][0]], g * 3)


def test_bf16_exchange_casts_and_one_launch_tap_pack():
    """The device side of the opt-in bf16 gradient exchange: t2v_cast_bf16 rounds like torch (round to nearest even, a NaN stays a
    NaN) for counts with a tail, and an arena with a bf16 exchange buffer packs / widens its [dense | live taps] halves in
    place (the collective itself is rehearsed over gloo in tests/test_dp_gloo.py)."""
    import torch
    from txt2vid_amd import functional as TF
    from txt2vid_amd.dist import GradArena
    dev = 'cuda:0'
    g = torch.Generator(device='cpu')
    g.manual_seed(3)
    x = (torch.randn(100003, generator=g) * 3).to(dev)
    x[5], x[6], x[7] = float('nan'), float('inf'), -0.0
    arena = GradArena([torch.nn.Parameter(torch.zeros(100003, device=dev))], TF.copy_into, exchange_dtype=torch.bfloat16)
    arena.flat.copy_(x)
    arena._cast(arena.flat, arena.half[:100003], True)
    want = x.to(torch.bfloat16)
    torch.cuda.synchronize()
    got = arena.half[:100003]
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    ok = ~torch.isnan(want)
    assert torch.equal(got[ok].view(torch.int16), want[ok].view(torch.int16))
    arena._cast(arena.half[:100003], arena.flat, False)
    torch.cuda.synchronize()
    assert torch.equal(arena.flat[ok], want[ok].float())
    # tap-major weights: the live rows go to the compact buffer (and back) in one multi-job launch per direction
    ws = [torch.nn.Parameter(TF.tap_major(torch.zeros(8, 6, 3, 3, device=dev))) for _ in range(9)]          # 9 jobs: two launches of <= 8
    b = torch.nn.Parameter(torch.zeros(5, device=dev))
    a2 = GradArena([b] + ws, TF.copy_into, live_taps={w: [4] for w in ws}, exchange_dtype=torch.bfloat16)
    vals = [torch.randn(8, 6, 3, 3, generator=g).to(dev) for _ in ws]
    for w, v in zip(ws, vals):
        w.grad = TF.tap_major(v.clone())
    b.grad = torch.ones(5, device=dev)
    a2.gather()
    a2._taps(True)
    torch.cuda.synchronize()
    for i, v in enumerate(vals):
        assert torch.equal(a2.compact[i * 48:(i + 1) * 48].view(8, 6), v[:, :, 1, 1])
    a2.compact.mul_(2.0)
    a2._taps(False)
    torch.cuda.synchronize()
    views = dict(zip([id(p) for p in a2.params], a2.views()))
    for w, v in zip(ws, vals):
        got = views[id(w)]
        assert torch.equal(got[:, :, 1, 1], 2 * v[:, :, 1, 1]) and torch.equal(got[:, :, 0, 0], v[:, :, 0, 0])
    assert a2.exchanged_bytes() == 2 * (5 + 9 * 48)
