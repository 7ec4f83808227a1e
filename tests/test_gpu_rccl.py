"""The per-step exchange on the RCCL backend itself (single rank: what a 1-GPU box can form), in a child process."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_exchange_primitives_run_on_the_rccl_backend(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    out = str(tmp_path / "rccl.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py"), port, out], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600, env=env)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
    res = json.load(open(out))
    assert res and all(res.values()), res
