"""stp3_amd -- MI355X-native implementation of ST-P3's LSS camera->BEV hot path.

The compute lives in hand-written HIP kernels (csrc/, exported through the C ABI declared in
include/stp3_hip.h); this package is the Python host that mirrors the reference's operator and
module surface (``stp3.models.stp3.STP3`` / ``stp3.trainer.TrainingModule``).
"""
__version__ = '0.1.0'
