#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q ${1:-} > gpurun_out/r03/gpu_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r03/gpu_tests.log
exit $rc
