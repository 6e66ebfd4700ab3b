#!/bin/bash
# A/B on the GPU box: bash scripts/r4_gpu_ab.sh TAG "pytest args or empty" variant...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=$1; shift; TESTS=$1; shift; O=gpurun_out/$TAG; mkdir -p $O
if [ -n "$TESTS" ]; then timeout 400 python -m pytest $TESTS -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log; fi
STEPS=${STEPS:-40} bash scripts/ab_variants.sh $TAG "$@"
