"""`loco_hd.atom_converter_utils` (/root/reference/loco_hd/atom_converter_utils.py): served by loco_hd_amd."""
from loco_hd_amd.atom_converter_utils import *  # noqa: F401,F403
from loco_hd_amd.atom_converter_utils import (PrimitiveAssigner, PrimitiveAtomSource, PrimitiveAtomTemplate, TypingSchemeElement)  # noqa: F401
