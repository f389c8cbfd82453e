"""C3 / C4 / C5 are built from the reference's own example models: tests/golden/topo_*.npz, parsed from
examples/models/{bulk_Im21,edl_Im21} by tests/golden/make_topologies.py.  Counts, thermostat partition and degrees of freedom as the
survey quotes them; the fixtures equal a fresh parse where the reference is present."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("openmm-velocityverlet_amd")
I, S = pkg.integrator, pkg.systems


def _plan(spec, **kw):
    it = I.VVIntegrator(333.0, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(0.02)
    plan, info, _ = I.create_plan(spec, it)
    pkg.vvhip.lib.vvhip_plan_destroy(plan)
    return info


def test_c3_is_bulk_im21_tiled_2x2x3():
    spec = S.make_config("C3", hbonds=True)
    assert spec.name == "bulk_Im21_2x2x3"
    assert (spec.num_atoms, spec.num_molecules, len(spec.drude_pairs), len(spec.constraints)) == (111000, 6000, 39000, 33000)   # SURVEY section 8 header
    assert np.allclose(spec.box, [6.2, 6.2, 18.3])
    # Drude = the particle right behind its parent, mass 0.4 (topol.psf)
    d, p = spec.drude_pairs[:, 0], spec.drude_pairs[:, 1]
    assert np.all(d == p + 1) and np.allclose(spec.masses[d], 0.4) and np.all(spec.mol_id[d] == spec.mol_id[p])
    # hydrogens are NOT next to their carbon in the real cation (the procedural look-alike of round 1 had them adjacent)
    h, x = spec.constraints[:, 0], spec.constraints[:, 1]
    assert np.allclose(spec.masses[h], 1.008) and (np.abs(h - x) > 1).mean() > 0.9
    assert np.allclose(np.linalg.norm(spec.positions[h] - spec.positions[x], axis=1), spec.constraint_distances, atol=1e-12)
    info = _plan(spec)
    assert np.allclose(list(info.dof), [165000.0, 17997.0, 117000.0], atol=1e-6)
    assert info.num_shake_clusters == 18000 and info.constraints_fused == 1
    assert info.num_waves == 1752 and info.num_slots_used == 111000                                    # 27 + 27 + 10 = 64: 99 % of the lanes
    assert S.make_config("C3x"[:2], 2.0).num_atoms == 222000


def test_c5_is_edl_im21():
    spec = S.make_config("C5")
    assert spec.name == "edl_Im21" and spec.num_atoms == 40310
    assert len(spec.particles_ld) == 2496 and len(spec.image_pairs) == 18907 and len(spec.particles_electrolyte) == 18907
    assert len(spec.drude_pairs) == 6643
    img = np.array(spec.image_pairs)
    assert np.all(spec.masses[img[:, 0]] == 0) and np.allclose(spec.charges[img[:, 0]], -spec.charges[img[:, 1]])
    assert np.all(spec.mol_id[img[:, 0]] == spec.mol_id[img[:, 1]])                                     # the zero bond of run-edl.py:95
    assert np.array_equal(img[:, 1], np.array(spec.particles_electrolyte))
    assert np.isclose(spec.box[2], 16.0)
    # images sit at the mirror position of their parents in conf.gro (mirror = Lz / 2); the image of a Drude particle was written at the
    # mirror position of the Drude's PARENT atom there, hence the 5e-3 nm
    z_i, z_p = spec.positions[img[:, 0], 2], spec.positions[img[:, 1], 2]
    assert np.abs(z_i - (16.0 - z_p)).max() < 6e-3
    info = _plan(spec)
    assert (info.num_particles_nh, info.num_molecules_nh, info.num_normal_ld, info.num_images) == (18907, 1022, 2496, 18907)
    assert info.num_temp_groups == 3 and info.use_com_temp_group == 1 and info.friction == 5.0
    hb = _plan(S.make_config("C5", hbonds=True))
    assert hb.num_shake_clusters == 511 * 6 and hb.constraints_fused == 1


def test_reduced_copies_keep_whole_ion_pairs():
    spec = S.make_config("C3", 0.004)
    assert spec.num_atoms == 12 * 37 and spec.num_molecules == 24 and len(spec.drude_pairs) == 12 * 13
    syn = S.make_config("C3", synthetic=True)
    assert syn.name.startswith("drude_il") and syn.num_atoms == 111000


@pytest.mark.skipif(not os.path.isdir("/root/reference/examples/models/bulk_Im21"), reason="reference only in the build container")
def test_fixtures_equal_a_fresh_parse(tmp_path):
    env = dict(os.environ)
    src = open(os.path.join(ROOT, "tests", "golden", "make_topologies.py")).read().replace("OUT = os.path.dirname(os.path.abspath(__file__))", f"OUT = {str(tmp_path)!r}")
    script = tmp_path / "mk.py"
    script.write_text(src)
    subprocess.run([sys.executable, str(script)], check=True, env=env, capture_output=True)
    for f in ("topo_bulk_Im21.npz", "topo_edl_Im21.npz"):
        a, b = np.load(tmp_path / f), np.load(os.path.join(ROOT, "tests", "golden", f))
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (f, k)
