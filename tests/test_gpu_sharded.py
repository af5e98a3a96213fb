"""-m gpu: the point-sharded path (SURVEY.md §8e way 2) with the HIP engine over a real RCCL process group.

What is reduced is the reference's own per-thread partial sum (/root/reference/include/nano_gicp/impl/nano_gicp_impl.hpp:260-267).
World size 1 is all a one-GPU box can host (RCCL refuses two ranks on one device); what it exercises is the part the two-handle
test in test_gpu_parity.py bypasses with a torch add between two device synchronisations: engine kernels -> all-reduce -> engine
kernels ordered by streams alone."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_sharded_align_over_rccl_world_size_1(hip_lib):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_sharded_rccl_worker.py"), str(_free_port()), "20000"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    print(out)
    assert out["backend"] == "nccl" and out["world_size"] == 1
    for r in out["runs"]:
        # one rank: the all-reduce is the identity, so the stepped alignment must reproduce align() to the last bit
        assert r["bit_equal"], r
        assert r["iters"][0] == r["iters"][1] and r["converged"][0] == r["converged"][1]
    # K1 sharded over the same group: the engine's buffer aliased by a torch tensor, block computed in place, committed
    assert out["covs"]["n"] == 20000 and out["covs"]["bit_equal"] and out["covs"]["alias_len"] == 6 * 20000
