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
