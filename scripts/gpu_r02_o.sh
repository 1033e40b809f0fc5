#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for d in 0 1 2 3 4; do LBC_GLDS_DIAG=$d timeout 300 python scripts/bench_ops.py 256 3 fwd l3.conv > $R/o_diag$d.log 2>&1; echo "== KT64 diag $d: $(grep l3.conv $R/o_diag$d.log)"; done
