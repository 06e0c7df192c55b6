"""CPU: `python bench.py --gpus N` without a launcher builds the driver's own torch.distributed.run command line and starts it as
a child before anything touches a GPU (Missing 1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_command_line(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '2'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    rc = bench.self_launch(4)
    cmd = seen['cmd']
    assert rc == 7                                                   # the ranks' exit code is passed on
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '2']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_bare_multi_gpu_invocation_goes_through_self_launch(monkeypatch):
    """main() with --gpus 2 and no WORLD_SIZE must not import the GPU stack: it exits with the child's return code."""
    sys.path.insert(0, ROOT)
    import bench
    import pytest
    monkeypatch.setattr(bench, 'self_launch', lambda n: 3 + n)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 5
