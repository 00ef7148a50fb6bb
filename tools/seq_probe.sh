#!/bin/bash
# kernels of the last batch of tools/seq_probe.py in launch order with durations.  usage: tools/seq_probe.sh MIN MAX [NEE]
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_sp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sp -- python3 $ROOT/tools/seq_probe.py "$@" > /tmp/sp.log 2>&1
python3 $ROOT/tools/trace_sequence.py "$(find /tmp/prof_sp -name '*kernel_trace.csv' | head -1)"
