#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "persistent loops:"; timeout -k 10 200 python3 tools/scst_train_time.py 2>&1 | grep "train step"
echo "per-step launches:"; COMIC_PERSIST=0 timeout -k 10 200 python3 tools/scst_train_time.py 2>&1 | grep "train step"
