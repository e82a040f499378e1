#!/bin/bash
# run_slack.sh <frames> <kind> <lib>...   (on the GPU box): alternating runs of library builds through tools/k1lab/slack.py
cd "$GRAFT_REPO_ROOT"
F=$1; K=$2; shift 2
for rep in 1 2; do for l in "$@"; do
  QOAUDIO_AMD_LIB=$PWD/tools/k1lab/bin/$l python tools/k1lab/slack.py $F $K 2>&1 | tail -1
done; done
