"""First `-m gpu` module (sorted before test_gpu_parity.py; this process has not touched the GPU yet): the RCCL backend and the
exchange step of the N > 1 bench, executed at world size 1 in a FRESH child process.

`bench.py --force-gather` initialises `backend="nccl"` (= RCCL on ROCm), decodes into bit-packed rows (written by the decode
kernels themselves, on the small and on the HBM-resident path), runs `dist.gather` of the device tensors through
`StepPipeline` with its nslots + 2 packed buffers and an `all_reduce` on a device tensor -- the lines of the N > 1 job that a
one-GPU box can execute before the driver's 8-GPU run does ("RCCL over xGMI only for the final gather")."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("extra,native", [(["--batch", "8192", "--steps", "3"], True),
                                          (["--config", "l29k_ms_e15", "--batch", "64", "--steps", "2"], True)])
def test_rccl_backend_and_device_gather_at_world_size_1(extra, native):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--warmup", "1",
           "--cpu-sample", "0", "--host-steps", "0"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    d = json.loads(lines[-1])
    assert d["value"] and d["value"] > 0
    assert d["gather_ms"] is not None and d["gather_ms"] >= 0.0
    assert d["gather"]["backend"].startswith("nccl") and d["gather"]["world_size"] == 1
    assert d["gather"]["native_packed_rows"] is native          # the kernels write packed rows themselves
    assert d["gather"]["packed_rows_equal_byte_rows"] is True    # ... and they are the byte rows, bit for bit
    assert d["gather"]["packed_buffers"] == d["config"]["pipelined_steps"] + 2
    assert d["cross_kernel_check"]["identical"] is True
    assert d["corrections_reproduce_syndromes"] is True
