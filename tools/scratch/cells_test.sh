#!/bin/bash
out=gpurun_out/cells; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_path.py -x -q -k "train_step_matches_oracle or greedy_and_beam_match_oracle" > $out/tests.log 2>&1
tail -25 $out/tests.log
