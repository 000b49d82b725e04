"""Structure -> primitive atoms -> scores on the GPU: PDB text in, scores out, against the oracle
chain (independent reader -> loop-for-loop assigner -> C oracle of the scoring path)."""
import io

import numpy as np
import pytest

from oracle import atom_converter_oracle as aco
from tests import pdb_util

pytestmark = pytest.mark.gpu

TIGHT = 1e-11


@pytest.fixture(scope="module")
def lh():
    import loco_hd_amd

    return loco_hd_amd


def _model(text, sid="s"):
    from loco_hd_amd.pdb_reader import PDBParser

    return PDBParser(QUIET=True).get_structure(sid, io.StringIO(text))[0]


def test_device_centroids_bit_exact(lh, tmp_path):
    """k_frames_centroids == np.mean(float32 members, axis=0) for every primitive atom of every frame, bit for bit."""
    from loco_hd_amd.device import DeviceSession

    pa = lh.PrimitiveAssigner(pdb_util.write_scheme(tmp_path / "scheme.json"))
    topo = pa.compile_topology(_model(pdb_util.synthetic_pdb(seed=21, n_res=60)))
    rng = np.random.default_rng(1)
    n_frames = 9
    frames = (topo.atom_coords[None] + rng.normal(0, 0.7, (n_frames,) + topo.atom_coords.shape)).astype(np.float32)
    single = next(p for p in range(len(topo)) if topo.src_start[p + 1] - topo.src_start[p] == 1)
    frames[2, topo.src_idx[topo.src_start[single]]] = -0.0  # a signed zero must survive a one-member mean
    lchd = lh.LoCoHD(pa.all_primitive_types)
    interner = {}
    packed = topo.pack(lchd, interner)
    sess = DeviceSession(lchd, interner=interner)
    ref = sess.upload(packed.xyz, packed.cat, packed.tag)
    buf = sess.frames_buffer(ref, n_frames)
    with pytest.raises(ValueError):
        sess.load_atom_frames(buf, frames)  # no source map yet
    sess.set_frame_sources(buf, topo)
    sess.load_atom_frames(buf, frames)
    got = sess.coords_of(buf, n_frames * len(topo)).reshape(n_frames, len(topo), 3)
    want = np.empty((n_frames, len(topo), 3), dtype=np.float32)
    for f in range(n_frames):
        for p in range(len(topo)):
            want[f, p] = np.mean([frames[f, i] for i in topo.src_idx[topo.src_start[p]:topo.src_start[p + 1]]], axis=0)
    assert got.dtype == np.float64 and np.array_equal(got.astype(np.float32), want) and np.array_equal(got, want.astype(np.float64))
    assert np.array_equal(np.signbit(got), np.signbit(want))
    # fewer frames than the capacity, and a bad map
    sess.load_atom_frames(buf, frames[:4])
    assert np.array_equal(sess.coords_of(buf, 4 * len(topo)).reshape(4, len(topo), 3), want[:4].astype(np.float64))
    bad = type(topo)(topo.primitive_types, topo.sources, topo.src_start, topo.src_idx + 10 ** 6, topo.n_atoms, topo.atom_coords)
    with pytest.raises(ValueError):
        sess.set_frame_sources(buf, bad)
    # source atoms already on the device, at every 4-byte misalignment of the frame block
    import torch

    sess.set_frame_sources(buf, topo)
    flat = torch.zeros(frames.size + 8, dtype=torch.float32, device="cuda")
    for shift in range(4):
        view = flat[shift:shift + frames.size].view(n_frames, topo.n_atoms, 3)
        view.copy_(torch.from_numpy(frames))
        sess.load_atom_frames_dev(buf, view)
        assert np.array_equal(sess.coords_of(buf, n_frames * len(topo)).reshape(n_frames, len(topo), 3), want.astype(np.float64)), shift
    sess.close()


def test_device_centroids_many_tiles_and_wide_groups(lh):
    """A map whose members span more source atoms than one LDS tile holds (several tiles per frame), primitive atoms made of
    far-apart atoms (global-gather tiles) and unsorted member lists, against np.mean."""
    from loco_hd_amd.atom_converter_utils import PrimitiveTopology
    from loco_hd_amd.device import DeviceSession

    rng = np.random.default_rng(7)
    n_atoms, n_prim = 20_000, 6_000
    start, idx = [0], []
    for p in range(n_prim):
        k = int(rng.integers(1, 9))
        if p % 997 == 0:      # spans (almost) the whole structure: cannot be staged
            members = rng.integers(0, n_atoms, 5)
        else:                 # local group, shuffled order
            lo = int(p * (n_atoms - 40) / n_prim)
            members = lo + rng.permutation(40)[:k]
        idx += [int(v) for v in members]
        start.append(len(idx))
    topo = PrimitiveTopology(["X"] * n_prim, [], np.asarray(start, np.int32), np.asarray(idx, np.int32), n_atoms,
                             rng.uniform(-50, 50, (n_atoms, 3)).astype(np.float32))
    n_frames = 3
    frames = (topo.atom_coords[None] + rng.normal(0, 1.0, (n_frames, n_atoms, 3))).astype(np.float32)
    lchd = lh.LoCoHD(["X"])
    sess = DeviceSession(lchd)
    ref = sess.upload(topo.centroids().astype(np.float64), np.zeros(n_prim, np.int32))
    buf = sess.frames_buffer(ref, n_frames)
    sess.set_frame_sources(buf, topo)
    sess.load_atom_frames(buf, frames)
    got = sess.coords_of(buf, n_frames * n_prim).reshape(n_frames, n_prim, 3)
    for f in range(n_frames):
        for p in list(range(0, n_prim, 53)) + [0, 997, n_prim - 1]:
            want = np.mean([frames[f, i] for i in idx[start[p]:start[p + 1]]], axis=0)
            assert np.array_equal(got[f, p], want.astype(np.float64)), (f, p)
    assert np.array_equal(got.astype(np.float32), topo.centroids(frames))
    sess.close()


def test_trajectory_from_source_atoms_matches_oracle(lh, oracle, tmp_path):
    """python_codes/trajectory_analyzer.py:97-119 with the per-frame conversion on the device: reference frame vs jittered
    frames, "Cent" anchors, accept_same=False, uniform[3,10], threshold 10."""
    from loco_hd_amd.device import DeviceSession

    scheme = pdb_util.write_scheme(tmp_path / "scheme.json")
    text = pdb_util.synthetic_pdb(seed=31, n_res=70, box=28.0)
    pa = lh.PrimitiveAssigner(scheme)
    topo = pa.compile_topology(_model(text))
    rng = np.random.default_rng(2)
    n_frames = 11
    frames = (topo.atom_coords[None] + rng.normal(0, 0.5, (n_frames,) + topo.atom_coords.shape)).astype(np.float32)
    cent = [i for i, t in enumerate(topo.primitive_types) if t == "Cent"]
    local_pairs = np.stack([cent, cent], 1)
    rule = {"accept_same": False}
    lchd = lh.LoCoHD(pa.all_primitive_types, lh.WeightFunction("uniform", [3.0, 10.0]), lh.TagPairingRule(rule))
    interner = {}
    packed = topo.pack(lchd, interner)
    sess = DeviceSession(lchd, interner=interner)
    ref = sess.upload(packed.xyz, packed.cat, packed.tag)
    got = sess.score_trajectory(ref, frames, local_pairs, 10.0, chunk=4, topology=topo)
    sess.close()
    assert got.shape == (n_frames, len(cent))
    # oracle chain: per frame, rebuild the residues with the frame's coordinates and run the loop-for-loop assigner
    sch = aco.load_scheme(scheme)
    residues0 = aco.read_pdb(text, "s")[0]

    def residues_of(frame):
        out, k = [], 0
        for full_id, resname, atoms in residues0:
            out.append((full_id, resname, [(nm, frame[k + j]) for j, (nm, _c) in enumerate(atoms)]))
            k += len(atoms)
        return out

    def prims(templates):
        return [oracle.PrimitiveAtom(t, f"{fid[2]}/{fid[3][1]}-{rn}", c) for t, c, (fid, rn, _n) in templates]

    lo = oracle.LoCoHD(pa.all_primitive_types, oracle.WeightFunction("uniform", [3.0, 10.0]), oracle.TagPairingRule(rule))
    ref_p = prims(aco.assign_primitive_structure(sch, residues0))
    for f in (0, 3, 4, 10):
        want = np.asarray(lo.from_primitives(ref_p, prims(aco.assign_primitive_structure(sch, residues_of(frames[f]))),
                                             [(int(i), int(i)) for i in cent], 10.0))
        assert np.max(np.abs(got[f] - want)) < TIGHT, f


@pytest.mark.parametrize("wfa", [None, {"function_name": "hyper_exp", "parameters": [1.0, 0.2]}])
def test_cli_end_to_end(lh, oracle, tmp_path, capsys, wfa):
    """`python -m loco_hd_amd` (loco_hd/__main__.py): two PDB files, a typing scheme, an anchor-pairing file -> output lines."""
    import json

    from loco_hd_amd import __main__ as cli

    scheme = pdb_util.write_scheme(tmp_path / "scheme.json")
    t1 = pdb_util.synthetic_pdb(seed=41, n_res=45, chains="A ", box=25.0, altlocs=False)   # chain " " like the README example
    t2 = pdb_util.synthetic_pdb(seed=42, n_res=45, chains="A ", box=25.0, altlocs=False)
    (tmp_path / "s1.pdb").write_text(t1)
    (tmp_path / "s2.pdb").write_text(t2)
    pa = lh.PrimitiveAssigner(scheme)
    p1, p2 = pa.assign_primitive_structure(_model(t1, "s1")), pa.assign_primitive_structure(_model(t2, "s2"))

    def ident(p):
        s = p.atom_source
        return f"{s.source_residue[2]}/{s.source_residue[3][1]}-{s.source_residue_name}/{','.join(s.source_atom)}"

    rng = np.random.default_rng(0)
    picks = [(int(a), int(b)) for a, b in zip(rng.integers(0, len(p1), 12), rng.integers(0, len(p2), 12))]
    entries = [f"{ident(p1[a])}:{ident(p2[b])}" for a, b in picks]
    pairing = ";\n".join(entries)  # newlines are stripped by the CLI, spaces are not
    (tmp_path / "pairs.txt").write_text(pairing)
    argv = ["-s1", str(tmp_path / "s1.pdb"), "-s2", str(tmp_path / "s2.pdb"), "-pts", str(scheme), "-apf", str(tmp_path / "pairs.txt")]
    if wfa is not None:
        argv += ["-wfa", json.dumps(wfa), "-udc", "12.5", "-tpra", '{"accept_same": true}']
    assert cli.main(argv) == 0
    got = capsys.readouterr().out.splitlines()
    want = aco.cli_lines(t1, t2, scheme, pairing, pa.all_primitive_types, cutoff=12.5 if wfa else 10.0,
                         tag_pairing_rule_args={"accept_same": True} if wfa else None, weight_function_args=wfa)
    assert len(got) == len(want) == len(entries)
    for g, w, e in zip(got, want, entries):
        assert g.startswith(f"LoCoHD({e}) = ") and w.startswith(f"LoCoHD({e}) = ")
        assert abs(float(g.split(" = ")[1]) - float(w.split(" = ")[1])) < TIGHT
    with pytest.raises(KeyError):  # an anchor that is not a primitive atom of the structure
        (tmp_path / "bad.txt").write_text("A/9999-GLY/CA:A/9999-GLY/CA")
        cli.main(argv[:6] + ["-apf", str(tmp_path / "bad.txt")])
