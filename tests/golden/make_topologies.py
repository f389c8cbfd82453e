#!/usr/bin/env python3
"""tests/golden/make_topologies.py -- run in the build container (needs /root/reference):

    python tests/golden/make_topologies.py

Parses the reference's example models into index / number fixtures (no file text is kept):

  topo_bulk_Im21.npz   <- examples/models/bulk_Im21/{conf.gro, topol.psf, ff.prm}
      masses, charges (topol.psf !NATOM columns 8, 7), mol_id (residue number: one ion = one bonded molecule), drude_pairs
      (Drude = atom name starting with "D", parent = the atom in front of it: examples/ommhelper/oplspsffile.py:1515),
      constraints + constraint_distances (constraints=HBonds, examples/run-bulk.py: every !NBOND bond with a hydrogen, at the b0 of
      its atom-type pair in ff.prm), positions (conf.gro, nm), box.
  topo_edl_Im21.npz    <- examples/models/edl_Im21/conf.gro  (the example ships no topol.psf for it)
      groups by residue name exactly as examples/run-edl.py:36-43: MoS2 -> Langevin subset, IMG -> images, everything else -> ionic
      liquid = electrolyte; image k mirrors ionic-liquid atom k (zip(group_ils, group_img)) and shares its molecule (the zero bond of
      run-edl.py:95); masses / charges / Drude pairs / HBonds of the ions from the bulk_Im21 topology by residue name and atom order,
      MoS2 by element (no Drudes there), images massless with the negated parent charge (run-edl.py:58-60).
"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/examples/models"
OUT = os.path.dirname(os.path.abspath(__file__))
ELEMENT_MASS = {"Mo": 95.95, "S": 32.06}


def read_gro(path):
    lines = open(path).read().splitlines()
    n = int(lines[1])
    resid, resname, name, xyz = [], [], [], []
    for ln in lines[2:2 + n]:
        resid.append(int(ln[0:5]))
        resname.append(ln[5:10].strip())
        name.append(ln[10:15].strip())
        xyz.append((float(ln[20:28]), float(ln[28:36]), float(ln[36:44])))
    box = np.array([float(x) for x in lines[2 + n].split()[:3]])
    return resid, resname, name, np.array(xyz), box


def read_psf(path):
    lines = open(path).read().splitlines()
    i = next(k for k, ln in enumerate(lines) if "!NATOM" in ln)
    n = int(lines[i].split()[0])
    atoms = []
    for ln in lines[i + 1:i + 1 + n]:
        f = ln.split()
        atoms.append(dict(resid=int(f[2]), resname=f[3], name=f[4], type=f[5], charge=float(f[6]), mass=float(f[7])))
    j = next(k for k, ln in enumerate(lines) if "!NBOND" in ln)
    nb = int(lines[j].split()[0])
    flat = []
    k = j + 1
    while len(flat) < 2 * nb:
        flat += [int(x) for x in lines[k].split()]
        k += 1
    bonds = np.array(flat, dtype=np.int64).reshape(-1, 2) - 1
    return atoms, bonds


def read_bond_lengths(path):
    """{(type1, type2): b0 [nm]} from the BONDS section of a CHARMM parameter file."""
    out, on = {}, False
    for ln in open(path):
        s = ln.strip()
        if s.startswith("BONDS"):
            on = True
            continue
        if on and re.match(r"^[A-Z]+\s*$", s) and not s.startswith("!"):
            break
        if on and s and not s.startswith("!"):
            f = s.split()
            if len(f) >= 4:
                out[(f[0], f[1])] = out[(f[1], f[0])] = float(f[3]) * 0.1
    return out


def hbonds(atoms, bonds, b0):
    cons, dist = [], []
    for a, b in bonds:
        ha, hb = atoms[a]["name"].startswith("H"), atoms[b]["name"].startswith("H")
        if ha == hb:
            continue
        h, x = (a, b) if ha else (b, a)
        cons.append((h, x))
        dist.append(b0[(atoms[h]["type"], atoms[x]["type"])])
    return np.array(cons, dtype=np.int32).reshape(-1, 2), np.array(dist)


def drude_pairs_of(names):
    d = np.array([i for i, nm in enumerate(names) if nm.startswith("D")], dtype=np.int32)
    return np.stack([d, d - 1], axis=1).astype(np.int32) if d.size else np.zeros((0, 2), np.int32)


def bulk():
    d = os.path.join(REF, "bulk_Im21")
    resid, resname, name, xyz, box = read_gro(os.path.join(d, "conf.gro"))
    atoms, bonds = read_psf(os.path.join(d, "topol.psf"))
    assert len(atoms) == len(name) == 9250
    b0 = read_bond_lengths(os.path.join(d, "ff.prm"))
    cons, dist = hbonds(atoms, bonds, b0)
    names = [a["name"] for a in atoms]
    mol = np.array([a["resid"] for a in atoms], dtype=np.int32) - 1
    assert np.all(np.diff(mol) >= 0) and mol.max() == 499
    pairs = drude_pairs_of(names)
    assert len(pairs) == 250 * 8 + 250 * 5 and len(cons) == 250 * 11
    np.savez_compressed(os.path.join(OUT, "topo_bulk_Im21.npz"), masses=np.array([a["mass"] for a in atoms]),
                        charges=np.array([a["charge"] for a in atoms]), mol_id=mol, drude_pairs=pairs, constraints=cons,
                        constraint_distances=dist, positions=xyz.astype(np.float32), box=box)
    # per-residue templates for the electrode slab, keyed by the (truncated) residue names of conf.gro
    tmpl = {}
    for key, full in (("c2c1i", "c2c1im"), ("dca", "dca")):
        first = next(i for i, a in enumerate(atoms) if a["resname"] == full)
        rid = atoms[first]["resid"]
        idx = [i for i, a in enumerate(atoms) if a["resid"] == rid]
        loc = {g: k for k, g in enumerate(idx)}
        tmpl[key] = dict(mass=np.array([atoms[i]["mass"] for i in idx]), charge=np.array([atoms[i]["charge"] for i in idx]),
                         names=[atoms[i]["name"] for i in idx],
                         cons=np.array([(loc[h], loc[x]) for h, x in cons if h in loc], dtype=np.int32).reshape(-1, 2),
                         dist=np.array([dd for (h, x), dd in zip(cons, dist) if h in loc]))
    return tmpl


def edl(tmpl):
    resid, resname, name, xyz, box = read_gro(os.path.join(REF, "edl_Im21", "conf.gro"))
    n = len(name)
    masses, charges, mol = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)
    group_mos = [i for i in range(n) if resname[i] == "MoS2"]
    group_img = [i for i in range(n) if resname[i] == "IMG"]
    group_ils = [i for i in range(n) if resname[i] not in ("MoS2", "IMG")]
    assert len(group_ils) == len(group_img) == 18907 and len(group_mos) == 2496
    cons, dist, nmol = [], [], 0
    i = 0
    while i < n:                                          # walk residue by residue
        j = i
        while j < n and resid[j] == resid[i] and resname[j] == resname[i]:
            j += 1
        rn = resname[i]
        if rn == "MoS2":                                  # one sheet = one molecule (no Drude particles in it)
            for k in range(i, j):
                masses[k] = ELEMENT_MASS[name[k]]
                mol[k] = nmol
            nmol += 1
        elif rn == "IMG":
            pass                                          # filled from the parents below
        else:
            t = tmpl[rn]
            assert j - i == len(t["mass"]), (rn, j - i)
            assert [x[0] for x in t["names"]] == [x[0] for x in name[i:j]], (rn, name[i:j])
            masses[i:j], charges[i:j], mol[i:j] = t["mass"], t["charge"], nmol
            for (h, x), dd in zip(t["cons"], t["dist"]):
                cons.append((i + h, i + x)); dist.append(dd)
            nmol += 1
        i = j
    image_pairs = np.array(list(zip(group_img, group_ils)), dtype=np.int32)      # (image, parent): addImagePair(image, parent), run-edl.py:93
    charges[image_pairs[:, 0]] = -charges[image_pairs[:, 1]]
    mol[image_pairs[:, 0]] = mol[image_pairs[:, 1]]                               # the zero bond of run-edl.py:95 joins them
    names_il = [nm if resname[k] not in ("MoS2", "IMG") else "X" for k, nm in enumerate(name)]
    pairs = drude_pairs_of(names_il)
    assert len(pairs) == 511 * 13
    np.savez_compressed(os.path.join(OUT, "topo_edl_Im21.npz"), masses=masses, charges=charges, mol_id=mol, drude_pairs=pairs,
                        constraints=np.array(cons, dtype=np.int32).reshape(-1, 2), constraint_distances=np.array(dist),
                        positions=xyz.astype(np.float32), box=box, particles_ld=np.array(group_mos, dtype=np.int32), image_pairs=image_pairs,
                        particles_electrolyte=np.array(group_ils, dtype=np.int32))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container)")
    edl(bulk())
    for f in ("topo_bulk_Im21.npz", "topo_edl_Im21.npz"):
        z = np.load(os.path.join(OUT, f))
        print(f, {k: z[k].shape for k in z.files}, os.path.getsize(os.path.join(OUT, f)), "bytes")
