"""Data-parallel step on hardware: two ranks (two processes) share cuda:0 and exchange gradients over gloo --
the statement of SURVEY.md 8e checked through GANStep itself (tests/dist_worker.py holds the assertions).
RCCL needs one device per rank, so the collective backend here is gloo; the trainer code path is the same."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_data_parallel_step_world2(tmp_path):
    world, port = 2, _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), str(r), str(world), str(port),
                               str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        path = tmp_path / ('rank%d.json' % r)
        assert path.exists(), 'rank %d wrote no result (rc %s):\n%s' % (r, p.returncode, outs[r][-3000:])
        res = json.loads(path.read_text())
        assert res['ok'] and p.returncode == 0, (res['fails'], outs[r][-2000:])
    # every rank saw its own batch: different losses
    r0, r1 = (json.loads((tmp_path / ('rank%d.json' % r)).read_text()) for r in range(2))
    assert r0['errD'] != r1['errD']
