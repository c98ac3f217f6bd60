"""MI355X-native implementation of the CoVA forward/backward hot path.

Host side mirrors the reference's ``models.CoVA`` / ``models.GraphAttentionLayer`` surface
(reference models.py:9-212); the device side is hand-written HIP for gfx950 behind the
C-ABI declared in ``include/cova_hip.h`` (``lib/libcova_hip.so``).  Import via
``import cova_amd`` from the repo root (the directory name is not a Python identifier).
"""
__all__ = ["weights", "synthetic"]
