#!/bin/bash
# A/B of library builds through tools/k1lab/classes.py: ab.sh <frames> <classes> <lib>...  (on the GPU box)
cd "$GRAFT_REPO_ROOT"
F=$1; C=$2; shift 2
for rep in 1 2; do for l in "$@"; do
  echo "== $l"; QOAUDIO_AMD_LIB=$PWD/tools/k1lab/bin/$l QOA_CLASSES=$C QOA_DEFAULT_ONLY=1 python tools/k1lab/classes.py $F 1024 2>&1 | grep "default"
done; done
