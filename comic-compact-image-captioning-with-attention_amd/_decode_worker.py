"""Worker side of inputs.DecodePool: JPEG / PNG files -> uint8 RGB written straight into a shared-memory staging block.

Runs in `spawn`ed processes that import only PIL and numpy -- never torch or the HIP library, so the workers are not
GPU processes (the loader's parallelism is then bounded by host cores, not by the per-GPU process limit) and do not
share the training process' interpreter lock (thread decode saturated at 2.3-2.9k images/s next to a step that
consumes 21k).  Counterpart of the `num_parallel_calls` map of the reference's tf.data pipeline
(common/inputs/manager_image_caption.py:163-175)."""
import numpy as np


def image_size(path):
    """(height, width) from the file header (no pixel decode)."""
    from PIL import Image
    with Image.open(path) as im:
        return int(im.size[1]), int(im.size[0])


def decode_into(args):
    """(path, shm name, byte offset, slot bytes) -> (height, width): the image is written at `offset` of the block when it
    fits its slot; (-height, -width) when it does not (nothing is written)."""
    from PIL import Image
    path, shm_name, off, slot = args
    with Image.open(path) as im:
        arr = np.asarray(im.convert('RGB'))
    if arr.size > slot:
        return -int(arr.shape[0]), -int(arr.shape[1])
    shm = _attach(shm_name)
    np.frombuffer(shm.buf, np.uint8, count=arr.size, offset=off)[:] = arr.reshape(-1)
    return int(arr.shape[0]), int(arr.shape[1])


_SHM = {}


def _attach(name):
    from multiprocessing import shared_memory
    s = _SHM.get(name)
    if s is None:
        s = _SHM[name] = shared_memory.SharedMemory(name=name)
    return s
