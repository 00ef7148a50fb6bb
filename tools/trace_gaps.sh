#!/bin/bash
# kernel busy time vs gaps of the last batches of a python command.  usage: tools/trace_gaps.sh script.py [args]
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_gap
S=$1; shift
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -- python3 $ROOT/$S "$@" > /tmp/gap.log 2>&1
python3 $ROOT/tools/trace_gaps.py "$(find /tmp/prof_gap -name '*kernel_trace.csv' | head -1)"
