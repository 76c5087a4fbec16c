#!/bin/bash
# round 5: the whole GPU suite on the current build (scale tests incl. the configs[3] / configs[4] stripes)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
T=${1:-r05suite}
O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp WTZ_TEST_KEEP_STDERR=$O/stderr
cd $R
( time timeout 3300 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
