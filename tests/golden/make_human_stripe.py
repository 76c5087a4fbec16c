#!/usr/bin/env python3
"""Parity pin for BASELINE configs[4]'s workload shape (synthetic 10 kb reads, 15 % error, 30x of a large genome, the human pipeline's `-k 17`:
smartdenovo.pl:16): the REAL reference `wtzmo -t 1 -k 17 -s 200 -m 0.6 -P 64 -p 0` - one stripe of the queries against the FULL index
(wtzmo.c:1291,1314) - on the seeded `human30` set of bench.py (30x of a 100 Mbp iid genome = 3 Gbp of reads, seed 59).  The full 90 Gbp job does not fit
this container in memory or time; the stripe pins the same code path at a size where the sharded k-mer index (wtzmo.c:1281-1303) and the per-batch
z-mer index are both in use.  Only checksums are committed (tests/golden/big_manifest.json, case "human30_zmo_P64p0"); the GPU test regenerates the reads
on the box (tests/test_gpu_scale.py, WTZ_TEST_HUMAN=1) and bench.py --workload human30 checks the same stripe as a second short pass.

Run in the build container only (oracle/_ref/wtzmo_ref):   python tests/golden/make_human_stripe.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_fly_stripe  # noqa: E402

if __name__ == "__main__":
    make_fly_stripe.main(shape="human", genome=100000000, coverage=30.0, seed=59, jobs_total=64, zmo=["-k", "17", "-s", "200", "-m", "0.6"], tmp="/tmp/wtz_human")
