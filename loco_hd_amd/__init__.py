"""loco_hd_amd -- MI355X-native drop-in for the scoring path of fazekaszs/loco_hd.

    from loco_hd_amd import LoCoHD, PrimitiveAtom, WeightFunction, TagPairingRule, StatisticalDistance

mirrors `from loco_hd import ...` (/root/reference/loco_hd/__init__.py:1); the Rust/PyO3 core is replaced
by libloco_hd_hip.so (C ABI: include/loco_hd_hip.h) which launches hand-written gfx950 kernels.
"""
from ._native import DeviceError, PanicException
from .api import LoCoHD, PrimitiveAtom, StatisticalDistance, TagPairingRule, WeightFunction
from .atom_converter_utils import (PrimitiveAssigner, PrimitiveAtomSource, PrimitiveAtomTemplate, PrimitiveTopology,
                                   TypingSchemeElement, prat_to_pra)

# `from loco_hd import *` gives the five core classes plus the converter classes (loco_hd/__init__.py:1-2)
__all__ = ["LoCoHD", "PrimitiveAtom", "StatisticalDistance", "TagPairingRule", "WeightFunction", "PanicException", "DeviceError",
           "PrimitiveAssigner", "PrimitiveAtomSource", "PrimitiveAtomTemplate", "TypingSchemeElement", "PrimitiveTopology",
           "prat_to_pra"]
