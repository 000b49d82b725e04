"""Loader for tests/golden/ (see tests/golden/make_golden.py for provenance)."""
import json
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"


def load_inputs(stamp):
    z = np.load(GOLD / f"ref_inputs_{stamp}.npz")
    meta = json.loads(bytes(z["meta"]).decode())
    offs = z["offsets"]
    seqs = [z["seq"][offs[k]:offs[k + 1]].astype(np.int32) for k in range(len(offs) - 1)]
    xyz = [z["xyz"][offs[k]:offs[k + 1]] for k in range(len(offs) - 1)]
    return meta, seqs, xyz


def load_cases():
    with open(GOLD / "oracle_outputs.json") as f:
        return json.load(f)["cases"]


def run_case(mod, case, cache={}):
    """Score one golden case with `mod` (oracle.oracle or loco_hd_amd); returns the score vector."""
    stamp = case["collection"]
    if stamp not in cache:
        cache[stamp] = load_inputs(stamp)
    meta, seqs, xyz = cache[stamp]
    types = meta["primitive_types"]
    sd = mod.StatisticalDistance(*(meta["statistical_distances"][case["sd"]] if case["sd"] is not None else ("Hellinger", [2.0])))
    lchd = mod.LoCoHD(types, mod.WeightFunction(*meta["weight_functions"][case["wf"]]), statistical_distance=sd)
    i, j = case["i"], case["j"]
    pa = [mod.PrimitiveAtom(types[s], "", c) for s, c in zip(seqs[i], xyz[i])]
    pb = [mod.PrimitiveAtom(types[s], "", c) for s, c in zip(seqs[j], xyz[j])]
    n = min(len(pa), len(pb))
    return np.asarray(lchd.from_primitives(pa, pb, [(x, x) for x in range(n)], meta["threshold_distance"]))


def stats(scores):
    return [float(np.mean(scores)), float(np.median(scores)), float(np.std(scores)), float(np.min(scores)), float(np.max(scores))]
