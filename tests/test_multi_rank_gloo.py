"""N > 1 path on CPU: two processes (gloo), each running its query stripe (-P 2 -p rank) through the host driver on the
emulated device layer, then the all_gather of the record blobs used by bench.py.  Each rank's output must equal the
reference golden of `wtzmo -t 1 -P 2 -p rank`, and every rank must see both blobs."""
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import GOLD, ROOT, case_argv, manifest

WORKER = r'''
import os, sys, hashlib, subprocess
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from smartdenovo_amd import multigpu
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
exe, inp, outdir = sys.argv[2], sys.argv[3], sys.argv[4]
argv = sys.argv[5:]
out = os.path.join(outdir, "r%d.ovl" % rank)
subprocess.run([exe, "-i", inp, "-fo", out] + argv + multigpu.stripe_argv(world, rank), check=True, capture_output=True)
dist.barrier()
blobs = multigpu.gather_records(dist, open(out, "rb").read(), "cpu")
open(os.path.join(outdir, "gathered_r%d.txt" % rank), "w").write(" ".join(hashlib.md5(b).hexdigest() for b in blobs))
dist.destroy_process_group()
'''


def test_two_rank_striping_and_gather(tmp_path):
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    exe = os.path.join(ROOT, "tests", "emul", "wtzmo_emul")
    m = manifest()["cases"]
    base = m["zmo"]
    w = os.path.join(str(tmp_path), "worker.py")
    open(w, "w").write(WORKER)
    import socket
    with socket.socket() as sk:      # a free port, not a fixed one (shared hosts)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, w, ROOT, exe, os.path.join(GOLD, base["input"]), str(tmp_path)] + case_argv(base), env=e))
    for p in procs:
        assert p.wait(timeout=900) == 0
    want = [m["zmo_P2p0"]["md5_full"], m["zmo_P2p1"]["md5_full"]]
    for r in range(2):
        got = hashlib.md5(open(os.path.join(str(tmp_path), "r%d.ovl" % r), "rb").read()).hexdigest()
        assert got == want[r], "rank %d stripe differs from reference -P 2 -p %d" % (r, r)
        assert open(os.path.join(str(tmp_path), "gathered_r%d.txt" % r)).read().split() == want
