"""hvqm4_amd -- MI355X-native HVQM4 1.3/1.5 picture reconstruction back end.

  sdk        Python mirror of the HVQM4 SDK C API (drop-in boundary, include/hvqm4.h)
  batch      batched device-resident path (include/hvqm4_amd.h)
  container  .h4m demux
  synth      synthetic stream writer (test / benchmark inputs)

Everything that touches pixels runs in libhvqm4_amd.so (HIP kernels for gfx950).
"""
__all__ = ["sdk", "batch", "container", "synth"]
