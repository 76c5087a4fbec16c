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
x.dev_is_host = True      # emulated device layer: the library's "device" buffers are host memory
x.install(host)
cargv = (C.c_char_p * (len(argv) + 1))(*[s.encode() for s in argv], None)
host.wtzmo_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
rc = host.wtzmo_main(len(argv), cargv)
open(os.path.join(outdir, "x%d.txt" % rank), "w").write("%d %d %d %d %d" % (rc, x.bytes_sent, x.bytes_received, x.messages, x.bytes_sent_from_device))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("name,world,extra", [("zmo", 2, []), ("dmo", 2, []), ("zmo", 3, []), ("zmo_n", 2, []), ("zmo", 2, ["--shard-index"]), ("dmo", 3, ["--shard-index"]),
                                              ("zmo", 2, ["--zindex-batch", "1"]), ("dmo", 3, ["--zindex-batch", "1"]), ("zmo", 2, ["--zindex-batch", "1", "--shard-index"]), ("zmo", 3, ["--zindex-batch", "1", "--shard-index"])])
def test_ranks_central_commit(name, world, extra, tmp_path):
    """extra = --shard-index: the k-mer index is sharded by read-id range over the RANKS (one (k-mer, count) exchange at build time, the
    groups of every query gathered from all ranks): rank 0's file must still be the unsharded `wtzmo -t 1` golden.
    --zindex-batch 1 (round 4: it used to exclude ranks): rank r rebuilds, per batch, the z-mer index of the batch's candidate reads = r (mod N) and of
    the batch's queries (WTZ_CMD_ZIDX) - with --shard-index on top this is the BASELINE configs[4] combination."""
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
    if not case["argv"].count("-U"):      # zmo: the CIGAR text - the bulk of what travels - left the ranks' DEVICE buffers (wtz_cigar_text_device), not their host copies
        assert all(s[4] > s[1] // 2 for s in stats[1:]), stats


def test_failed_rank_ends_all_ranks_in_band(tmp_path):
    """A rank that fails reports it in the status word of its next reply; rank 0 finishes the round of the exchange, broadcasts WTZ_CMD_ABORT and every
    process exits 1 (the reference's convention: every error is exit(1), list.h:64-67) - nobody stays blocked in a receive until a launcher kills it."""
    subprocess.run([os.path.join(ROOT, "tests", "emul", "build_emul.sh")], check=True)
    lib = os.path.join(ROOT, "tests", "emul", "libwtzmo_host_emul.so")
    case = manifest()["cases"]["zmo"]
    w = os.path.join(str(tmp_path), "worker.py")
    open(w, "w").write(WORKER)
    for failing in (1, 0):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WTZ_RANK_FAIL_AT="%d:2" % failing)
        procs = []
        for r in range(3):
            e = dict(env, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, w, ROOT, lib, os.path.join(GOLD, case["input"]), str(tmp_path)] + case_argv(case), env=e, stderr=subprocess.PIPE))
        errs = []
        for p in procs:
            _, err = p.communicate(timeout=300)          # a hang here is the failure this test exists for
            errs.append(err.decode(errors="replace"))
        assert [p.returncode for p in procs] == [1, 1, 1], [p.returncode for p in procs]
        assert "ending all 3 ranks" in errs[0] and ("rank %d failed" % failing) in errs[0]
        for r in (1, 2):
            assert "abort requested by rank 0" in errs[r]
