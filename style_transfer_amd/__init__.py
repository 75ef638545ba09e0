"""MI355X-native tiled neural style transfer (the hot path of crowsonkb/style_transfer).

Host side in Python mirrors the reference's tile-worker interface; all arithmetic runs in
hand-written HIP kernels for gfx950 behind the C ABI in ``include/stx.h``
(``style_transfer_amd/csrc``).  There is no CPU fallback: importing ``style_transfer_amd.lib``
without the built ``libstx.so`` raises.
"""

__version__ = '0.1.0'
