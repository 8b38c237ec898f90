#!/bin/bash
cd $GRAFT_REPO_ROOT
export M=25 C=2048 CG=2048 B=64 N=30
echo "MODE 0:"; COMIC_PERSIST_STAMPS=1 timeout -k 10 200 python3 tools/dec_step_time.py 2>&1 | grep -E "persist stamps bwd|decoder step" | tail -2
echo "own rows (MODE 2):"; COMIC_BWD_OWN_ROWS=1 COMIC_PERSIST_STAMPS=1 timeout -k 10 200 python3 tools/dec_step_time.py 2>&1 | grep -E "persist stamps bwd|decoder step" | tail -2
COMIC_BWD_OWN_ROWS=1 timeout -k 10 400 python -m pytest tests/test_gpu_path.py -q -m gpu -x -k "decoder_train_step_matches_oracle or persistent_loops" 2>&1 | tail -2
