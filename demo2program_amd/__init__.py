"""demo2program full-model training step, MI355X-native (gfx950 HIP kernels behind a C ABI).

Layout:
  csrc/              HIP kernels + the C ABI declared in include/d2p.h
  lib.py             ctypes binding (fails loudly when the library is missing)
  kernels.py         thin tensor-level wrappers over the C ABI
  models/model_full.py, trainer.py
                     host-side mirror of the reference surface
                     (models/model_full.py:22, trainer.py:16 in shaohua0116/demo2program)
"""
__version__ = '0.1.0'
