#!/bin/bash
# tools/grun.sh [--timeout S] -- '<command>': gpurun with the commit of the tree stamped into build_commit.txt first (the GPU box gets no .git;
# tools/pmc_reduce.py and bench.py record it with what they measure).  A tree with uncommitted changes is stamped <commit>-dirty.
cd "$(dirname "$0")/.." || exit 1
c=$(git rev-parse --short HEAD)
if ! git diff --quiet HEAD -- . ':!profiles' ':!*.md'; then c="$c-dirty"; fi
echo "$c" > build_commit.txt
exec /usr/local/graft/bin/gpurun "$@"
