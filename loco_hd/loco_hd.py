"""`loco_hd.loco_hd` -- the name of the reference's extension module (/root/reference/src/lib.rs:9-17): its five classes."""
from loco_hd_amd import LoCoHD, PrimitiveAtom, StatisticalDistance, TagPairingRule, WeightFunction

__all__ = ["LoCoHD", "PrimitiveAtom", "StatisticalDistance", "TagPairingRule", "WeightFunction"]
