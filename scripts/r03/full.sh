#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/full; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?; tail -8 $O/tests.log; [ $rc = 0 ] || exit $rc
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
