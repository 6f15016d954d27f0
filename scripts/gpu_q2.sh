#!/bin/bash
TAG=${1:-q}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
python scripts/gpu_ctcprof.py 2>&1 | tail -4
bash scripts/gpu_quick.sh $TAG "$@"
