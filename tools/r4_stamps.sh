#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in "1280,12,12,768,192,1,1,1" "1280,12,12,192,192,7,1,1"; do
echo "== $c"; CASE=$c PAD=SAME TILE=44,28 timeout -k 10 120 python3 tools/stamps_dma.py 2>&1 | grep -v amdgpu
done
