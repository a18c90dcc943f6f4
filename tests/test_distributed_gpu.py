"""N>1 path on a GPU box: two ranks share cuda:0 (bench.py --single-device, gloo rendezvous on 127.0.0.1) and run
the full step -- sharded level 0, exchange of the W slabs, level 1 with tile-sharded Gram / system-sharded
solves completed by the all-reduce callback.  The LOCO checksum and the selected tau must equal the
single-process run of the same problem (weak-scaling bench: 2 x 1500 SNPs == 1 x 3000 SNPs)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(out):
    for ln in reversed(out.strip().split("\n")):
        if ln.startswith("{"):
            return json.loads(ln)
    raise AssertionError(out)


@pytest.mark.parametrize("phenos", [1, 4])
def test_two_ranks_equal_one_rank(phenos):
    """phenos = 1: W all-gathered, level 1 shared tile-wise (all-reduce callback).  phenos = 4 (>= world): predictor
    slabs exchanged by phenotype (all-to-all), each rank runs level 1 for its own two phenotypes (rg_set_l1_view)."""
    common = ["--samples", "2048", "--bsize", "200", "--steps", "1", "--warmup", "0", "--no-cpu", "--phenos", str(phenos)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--snps", "3000"] + common,
                        capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    one = _line(r1.stdout)
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                         os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo",
                         "--snps", "1500"] + common, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    two = _line(r2.stdout)
    assert two["n_gpus"] == 2 and one["config"]["snps"] == two["config"]["snps"] == 3000
    assert one["selected_tau_index"] == two["selected_tau_index"]
    assert abs(one["loco_checksum"] - two["loco_checksum"]) <= 1e-9 * abs(one["loco_checksum"])
