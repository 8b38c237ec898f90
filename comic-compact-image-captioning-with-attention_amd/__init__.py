"""comic_amd -- MI355X-native hot path of COMIC image captioning.

Host side (Python) mirrors the reference's operator interface for the path
CNN encoder -> attention-LSTM decoder -> SCST loop; all arithmetic runs in the
hand-written HIP library `lib/libcomic_hip.so` through the C-ABI declared in
`include/comic_hip.h`.  There is NO CPU fallback: importing a compute module without
the built library raises.
"""
import os as _os

# The training executors keep up to six streams busy at once (main, the two chain lanes and the weight-gradient lane of the CNN
# backward, the encoder's side stream, the all-reduce stream; the inference loop two decode lanes).  ROCm maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin in creation order, and streams that share a queue run one after
# the other: with two unrelated streams created first, the cnn_finetune step took 6.78 ms instead of 5.33 (round 5).  Ask for
# eight queues unless the user has chosen; read by the HIP runtime when it initialises, so this must run before the first GPU
# call of the process (bench.py, src/train.py, src/infer.py and tests/conftest.py import this package / set it first).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'
