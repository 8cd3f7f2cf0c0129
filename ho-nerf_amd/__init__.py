"""MI355X-native volume-rendering core behind HO-NeRF's renderer call surface.

Import as ``honerf_amd`` (see ../honerf_amd/__init__.py).  Submodules:
  synth      synthetic weights / cameras / poses (numpy only)
  lib        ctypes binding of the C-ABI library (include/honerf.h)
  nets       parameter containers in the reference's state-dict layout
  renderer   NeuSRenderer / NeuSRenderer_fitting adapters (utils/renderer.py surface)
  renderer_batch  frame-batched NeuSRenderer_fitting (utils/renderer_batch.py surface)
  fitting    frame-sharded fitting drivers
"""
__version__ = '0.1.0'
