"""`loco_hd` -- the import name the reference's callers use, served by the MI355X-native build.

    from loco_hd import LoCoHD, PrimitiveAtom, WeightFunction, TagPairingRule, StatisticalDistance

is what every caller of fazekaszs/loco_hd writes (/root/reference/loco_hd/__init__.py:1-2,
python_codes/simple_test.py:4, README.md:373-390).  This package only re-exports `loco_hd_amd`; with the repository root on
`sys.path` (or the two packages installed side by side) such a caller runs unchanged on the HIP path.  Do not install it next to
the reference's own `loco_hd` wheel: the two would shadow each other.
"""
from loco_hd_amd import (DeviceError, LoCoHD, PanicException, PrimitiveAssigner, PrimitiveAtom, PrimitiveAtomSource,
                         PrimitiveAtomTemplate, PrimitiveTopology, StatisticalDistance, TagPairingRule, TypingSchemeElement,
                         WeightFunction, prat_to_pra)

__all__ = ["LoCoHD", "PrimitiveAtom", "StatisticalDistance", "TagPairingRule", "WeightFunction", "PanicException", "DeviceError",
           "PrimitiveAssigner", "PrimitiveAtomSource", "PrimitiveAtomTemplate", "TypingSchemeElement", "PrimitiveTopology",
           "prat_to_pra"]
