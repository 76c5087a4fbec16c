"""N > 1 path on CPU: world_size 2 and 3 (gloo), every rank running the host driver (the C `wtzmo` main as a shared object) on the
emulated device layer with the exchange hooks of smartdenovo_amd/multigpu.py.  Rank 0 plans and commits, all ranks compute their
share of the pairs and of the candidate requests: the ONE .ovl rank 0 writes must be the reference's plain `wtzmo -t 1` golden
(not the union of -P stripes), the other ranks write nothing, and records must really have travelled."""
import hashlib
import os
import socket
import subprocess
import sys

import pytest

from conftest import GOLD, ROOT, case_argv, manifest

WORKER = r'''
import os, sys, ctypes as C
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from smartdenovo_amd import multigpu
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
lib, inp, outdir = sys.argv[2], sys.argv[3], sys.argv[4]
argv = ["wtzmo", "-i", inp, "-fo", os.path.join(outdir, "r%d.ovl" % rank), "--batch", "16"] + sys.argv[5:]
host = C.CDLL(lib)
x = multigpu.RankExchange(dist, "cpu")
x.install(host)
cargv = (C.c_char_p * (len(argv) + 1))(*[s.encode() for s in argv], None)
host.wtzmo_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
rc = host.wtzmo_main(len(argv), cargv)
open(os.path.join(outdir, "x%d.txt" % rank), "w").write("%d %d %d %d" % (rc, x.bytes_sent, x.bytes_received, x.messages))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("name,world,extra", [("zmo", 2, []), ("dmo", 2, []), ("zmo", 3, []), ("zmo_n", 2, []), ("zmo", 2, ["--shard-index"]), ("dmo", 3, ["--shard-index"])])
def test_ranks_central_commit(name, world, extra, tmp_path):
    """extra = --shard-index: the k-mer index is sharded by read-id range over the RANKS (one (k-mer, count) exchange at build time, the
    groups of every query gathered from all ranks): rank 0's file must still be the unsharded `wtzmo -t 1` golden."""
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    lib = os.path.join(ROOT, "tests", "emul", "libwtzmo_host_emul.so")
    case = manifest()["cases"][name]
    w = os.path.join(str(tmp_path), "worker.py")
    open(w, "w").write(WORKER)
    with socket.socket() as sk:      # a free port, not a fixed one (shared hosts)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, w, ROOT, lib, os.path.join(GOLD, case["input"]), str(tmp_path)] + case_argv(case) + extra, env=e))
    for p in procs:
        assert p.wait(timeout=900) == 0
    got = hashlib.md5(open(os.path.join(str(tmp_path), "r0.ovl"), "rb").read()).hexdigest()
    assert got == case["md5_full"], "rank 0's .ovl differs from reference wtzmo -t 1"
    assert hashlib.md5(open(os.path.join(str(tmp_path), "r0.ovl.contained"), "rb").read()).hexdigest() == case["md5_contained"]
    stats = [[int(v) for v in open(os.path.join(str(tmp_path), "x%d.txt" % r)).read().split()] for r in range(world)]
    assert all(s[0] == 0 for s in stats)
    for r in range(1, world):
        assert os.path.getsize(os.path.join(str(tmp_path), "r%d.ovl" % r)) == 0, "only rank 0 writes records"
        assert stats[r][1] > 1000, "rank %d computed nothing" % r       # bytes it sent back to rank 0
    assert stats[0][2] == sum(s[1] for s in stats[1:]) and stats[0][2] > 0
