#!/bin/bash
# dmo parity + timing on the repeat-rich set and E. coli shape after a dot-matrix change
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R; export TMPDIR=/tmp
bash tools/gpu_r02_rep.sh
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or fresh" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_scale.py -q -x -k "repeat or ecoli" 2>&1 | tail -3
