"""Import shim for the package directory ``cova-web-object-detection_amd/``.

The directory name carries hyphens (it is derived from the reference repository's
name), which Python's ``import`` statement cannot spell.  This module registers the
directory as the importable package ``cova_web_object_detection_amd`` so that::

    import cova_amd                                  # registers the package
    from cova_web_object_detection_amd.models import CoVA, GraphAttentionLayer

works from the repo root (tests, bench.py and __graft_entry__.py all go through it).
"""
import importlib.util
import os
import sys

PKG_NAME = "cova_web_object_detection_amd"
PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cova-web-object-detection_amd")


def load():
    if PKG_NAME in sys.modules:
        return sys.modules[PKG_NAME]
    spec = importlib.util.spec_from_file_location(
        PKG_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = load()
