#!/bin/bash
# usage (on the GPU box, from the repo root): profiles/rocprof_run.sh <out_summary.txt> <steps-divisor> -- <command...>
# rocprofv3 --kernel-trace --stats of <command>, summarised per kernel and per (kernel, grid) by summarize_rocpd.py
out=$1; steps=$2; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tdr_prof
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d /tmp/tdr_prof -o t -- "$@" > /tmp/tdr_prof_cmd.log 2>&1
db=$(find /tmp/tdr_prof -name '*.db' | head -1)
cd "$root" && python profiles/summarize_rocpd.py "$db" "$steps" > "$out" 2>&1
