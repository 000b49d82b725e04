#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/ (run in the build container; needs /root/reference).

INPUTS  come from the reference's own test data, the two input collections
        /root/reference/tests/test_data/test_input_collection_{241212_131752,250605_114749}.pickle
        (random labelled clouds, weight-function and statistical-distance parameters drawn by
        /root/reference/tests/generate_locohd_testcases.py).  They are repacked as .npz (data only).
OUTPUTS the reference's matching output pickles are NOT in the tree (.MISSING_LARGE_BLOBS) and the Rust core
        cannot be built here, so the expected values are produced by the CPU oracle (oracle/locohd_oracle.c,
        itself pinned to the reference's hand-computed known answers in tests/test_oracle_kat.py).  They are
        labelled "oracle" in the JSON: a regression net for the oracle and the parity target for the HIP path,
        not an independent confirmation of the reference.

Each case mirrors the reference's consistency test (/root/reference/tests/test_locohd.py:75-133):
from_primitives(cloud i, cloud j, anchors (x, x) for x < min(len), threshold 50) -> (mean, median, std, min, max).
"""
import json
import pickle
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
REF = Path("/root/reference/tests/test_data")


def repack(stamp):
    with open(REF / f"test_input_collection_{stamp}.pickle", "rb") as f:
        col = pickle.load(f)
    types = list(col["primitive_types"])
    seqs = [np.asarray([types.index(s) for s in scp["seq"]], dtype=np.int8) for scp in col["sequence_coordinate_pairs"]]
    xyz = [np.asarray(scp["coords"], dtype=np.float64) for scp in col["sequence_coordinate_pairs"]]
    offs = np.cumsum([0] + [len(s) for s in seqs])
    meta = {"primitive_types": types, "threshold_distance": float(col["threshold_distance"]),
            "weight_functions": [[n, list(map(float, p))] for n, p in col["weight_functions"]],
            "statistical_distances": [[n, list(map(float, p))] for n, p in col.get("statistical_distances", [])]}
    np.savez_compressed(HERE / f"ref_inputs_{stamp}.npz", seq=np.concatenate(seqs), xyz=np.concatenate(xyz), offsets=offs,
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    return meta, seqs, xyz


def cases_for(stamp, meta, n_clouds):
    out = []
    if stamp.startswith("241212"):  # every weight function, Hellinger-2, one cloud pair each
        for w in range(len(meta["weight_functions"])):
            out.append((w, None, w % n_clouds, (w + 1) % n_clouds))
    else:  # every (weight function, statistical distance) combination
        for w in range(len(meta["weight_functions"])):
            for d in range(len(meta["statistical_distances"])):
                k = w * len(meta["statistical_distances"]) + d
                out.append((w, d, k % n_clouds, (k + 3) % n_clouds))
    return out


def main():
    from oracle import oracle as orc

    expected = {"generator": "oracle (oracle/locohd_oracle.c)", "cases": []}
    for stamp in ("241212_131752", "250605_114749"):
        meta, seqs, xyz = repack(stamp)
        types = meta["primitive_types"]
        for w, d, i, j in cases_for(stamp, meta, len(seqs)):
            sd = orc.StatisticalDistance(*(meta["statistical_distances"][d] if d is not None else ("Hellinger", [2.0])))
            lchd = orc.LoCoHD(types, orc.WeightFunction(*meta["weight_functions"][w]), statistical_distance=sd)
            n = min(len(seqs[i]), len(seqs[j]))
            tag = lambda m: np.zeros(m, dtype=np.int32)
            scores = np.asarray(lchd.from_arrays(xyz[i], seqs[i].astype(np.int32), tag(len(seqs[i])), xyz[j],
                                                 seqs[j].astype(np.int32), tag(len(seqs[j])), [(x, x) for x in range(n)],
                                                 meta["threshold_distance"]))
            rec = {"collection": stamp, "wf": w, "sd": d, "i": i, "j": j,
                   "stats": [float(np.mean(scores)), float(np.median(scores)), float(np.std(scores)), float(np.min(scores)),
                             float(np.max(scores))]}
            if len(expected["cases"]) % 37 == 0:
                rec["scores"] = scores.tolist()
            expected["cases"].append(rec)
    with open(HERE / "oracle_outputs.json", "w") as f:
        json.dump(expected, f)
    print(len(expected["cases"]), "cases")


if __name__ == "__main__":
    main()
