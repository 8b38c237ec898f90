"""comic_amd -- MI355X-native hot path of COMIC image captioning.

Host side (Python) mirrors the reference's operator interface for the path
CNN encoder -> attention-LSTM decoder -> SCST loop; all arithmetic runs in the
hand-written HIP library `lib/libcomic_hip.so` through the C-ABI declared in
`include/comic_hip.h`.  There is NO CPU fallback: importing a compute module without
the built library raises.
"""
__version__ = '0.1.0'
