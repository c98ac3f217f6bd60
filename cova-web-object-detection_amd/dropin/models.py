"""Drop-in replacement for the reference's ``models.py``: put this directory (and the repo root)
ahead of the reference on PYTHONPATH and ``from models import CoVA`` (reference main.py:10,
evaluate.py:9, extract_attn_wts_and_visualize.py:10) resolves to the MI355X implementation."""
import cova_amd  # noqa: F401  (registers cova_web_object_detection_amd)
from cova_web_object_detection_amd.models import CoVA, GraphAttentionLayer  # noqa: F401
