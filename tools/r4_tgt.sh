#!/bin/bash
cd $GRAFT_REPO_ROOT
export M=25 C=2048 CG=2048 B=64 N=30
for t in 800 1000 1200 1600 2000; do for f in 5; do echo -n "target $t flags $f: "; GG_TARGET=$t GG_FLAGS=$f timeout -k 10 300 python tools/dec_step_time.py 2>&1 | tail -1; done; done
