"""Import alias: ``import susnet_amd`` == the package in ``sus-net_amd/`` (a hyphen is not importable)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("sus-net_amd")
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("sus-net_amd."):
        sys.modules["susnet_amd." + _name.split(".", 1)[1]] = _mod
sys.modules[__name__] = _pkg
