"""realvsr_amd -- MI355X (gfx950) implementation of RealVSR's EDVR alignment / fusion /
reconstruction / pyramid-loss hot path behind the reference's own operator and arch API.

Layout (mirrors the reference modules the path touches):
    realvsr_amd.archs.dcn          <-> codes/models/archs/dcn          (operator boundary)
    realvsr_amd.archs.EDVR_arch    <-> codes/models/archs/EDVR_arch.py
    realvsr_amd.archs.arch_util    <-> codes/models/archs/arch_util.py
    realvsr_amd.VideoSR_archs      <-> codes/models/VideoSR_archs.py   (define_G)
    realvsr_amd.loss               <-> codes/models/loss.py
    realvsr_amd.util               <-> codes/utils/util.py             (pyramid helpers)
    realvsr_amd.csrc               HIP kernels + C ABI (include/realvsr_hip.h)
All compute runs in librealvsr_hip.so; there is no CPU or eager-PyTorch fallback.
"""
__version__ = '0.1.0'
