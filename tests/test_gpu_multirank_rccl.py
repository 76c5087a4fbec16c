"""The RCCL carrier of the one-process-per-GPU form (what shards: wtzmo.c:1291,1314) on REAL devices: `bench.py --gpus 2` on the nccl backend -
rank 0 plans and commits, rank 1 serves pair / candidate requests, results come back with send / recv, the CIGAR text from device memory through a tensor that
aliases the library's buffer (smartdenovo_amd/multigpu.py::_send_dev).  The build box and the round's GPU box have ONE device, so this test skips itself there
(the gloo tests of tests/test_multi_rank_gloo.py cover the protocol on the CPU); on any box with two it runs the configs[1] workload for one step and checks
the three things that path has never been able to prove: two ranks ran, the .ovl is the reference's (md5 of `wtzmo -t 1`), bytes left a peer's DEVICE memory."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _device_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


@pytest.mark.parametrize("extra", [[], ["--engine", "dmo"]], ids=["zmo", "dmo"])
def test_two_ranks_over_rccl_write_the_reference_file(extra):
    if _device_count() < 2:
        pytest.skip("fewer than two visible GPUs: RCCL needs one device per rank")
    env = dict(os.environ, WTZ_BENCH_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "ecoli", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + extra,
                       capture_output=True, env=env, timeout=1800)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode().strip().split("\n")[-1])
    assert line["n_gpus"] == 2 and line["backend"] == "nccl"
    assert line["parity"]["match"] is True, line["parity"]
    if not extra:       # the zmo records carry CIGAR text: it must have left rank 1 from device memory
        assert line["bytes_sent_from_device"] > 0
    assert line["gathered_result_bytes_per_step"] > 0 and line["exchange_messages_per_step"] > 0
