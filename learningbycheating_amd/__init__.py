"""MI355X-native implementation of the LearningByCheating sensorimotor training hot path.

Drop-in boundary = the reference's module API:
    from learningbycheating_amd.bird_view.models.image import ImagePolicyModelSS
    from learningbycheating_amd.bird_view.models.birdview import BirdViewPolicyModelSS
Per-step compute runs in hand-written HIP kernels for gfx950 (csrc/), reached through the
C ABI of include/lbc_hip.h; torch only provides device memory, streams and RCCL.
"""
__version__ = "0.1.0"
