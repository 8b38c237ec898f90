set -e
mkdir -p gpurun_out/r2_exp1
O=gpurun_out/r2_exp1
# 1x1 768->192 at 12x12, B=640 (input 141 MB) vs B=128 (28 MB: L2-resident)
CASE=640,12,12,768,192,1,1,1 TILE=44,41,35,40,27,28,30,31,1 python scratch/one_conv.py > $O/c1x1_640.log 2>&1
CASE=128,12,12,768,192,1,1,1 TILE=44,41,35,40,27,28,30,31,1 python scratch/one_conv.py > $O/c1x1_128.log 2>&1
CASE=640,12,12,192,192,7,1,1 PAD=SAME TILE=44,41,35,40,27,28 python scratch/one_conv.py > $O/c7x1_640.log 2>&1
CASE=640,12,12,768,192,1,1,1 TILE=44,41,27 python scratch/stamps_dma.py > $O/st1x1_640.log 2>&1
CASE=640,12,12,192,192,7,1,1 PAD=SAME TILE=44,41,27 python scratch/stamps_dma.py > $O/st7x1_640.log 2>&1
tail -n 40 $O/*.log
