"""SURVEY.md section 8 row f2, turnkey: on the first machine that has OpenMM >= 8.2 with its HIP platform, swig and cmake, ONE command

    python -m pytest tests/test_openmm_dropin.py -m gpu -q

builds this repository against that OpenMM (CMakeLists.txt -DOPENMM_DIR=..., the SWIG module of python/velocityverletplugin.i), loads the
plugin through OpenMM's own plugin loader and steps the C3 box through `openmm.Context(..., Platform.getPlatformByName('HIP'))` with
`velocityverletplugin.VVIntegrator` exactly as examples/run-bulk.py:56-79 does -- platform name and precision property being the only
differences a user of the reference would type (`'HIP'` / `'HipPrecision'` for `'CUDA'` / `'CudaPrecision'`).  The end state is held to the
reference pipeline's recorded one (tests/golden/refhost_C3_full.npz: the reference's own VVIntegrator.cpp + CudaVVKernels.cpp + kernels,
4 steps, static forces).

Nothing of this can run in the development image (no OpenMM, no swig, no network): every prerequisite that is missing SKIPS the test and
says which.  It is kept small on purpose; its value is that row f2 gets answered the day the prerequisites exist.
"""
import importlib
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu


def _prerequisites():
    try:
        import openmm as mm
    except Exception as e:                                   # noqa: BLE001
        return None, f"import openmm: {type(e).__name__}: {e}"
    names = [mm.Platform.getPlatform(i).getName() for i in range(mm.Platform.getNumPlatforms())]
    if "HIP" not in names:
        return None, f"this OpenMM ({mm.version.version}) has no HIP platform (platforms: {names})"
    for tool in ("cmake", "swig"):
        if shutil.which(tool) is None:
            return None, f"{tool} is not on the PATH"
    lib = mm.version.openmm_library_path                     # <prefix>/lib
    prefix = os.path.dirname(lib)
    if not os.path.isfile(os.path.join(prefix, "include", "openmm", "hip", "HipContext.h")):
        return None, f"{prefix}/include/openmm/hip/HipContext.h not found: the HIP platform's headers are needed to build the kernel factory"
    return (mm, prefix), ""


def _build(prefix, work):
    """cmake + swig + setup.py, all into `work` (nothing is installed into the OpenMM tree)."""
    vvhip = os.path.join(ROOT, "openmm-velocityverlet_amd", "lib", "libvvhip.so")
    subprocess.run(["cmake", "-S", ROOT, "-B", os.path.join(work, "build"), f"-DOPENMM_DIR={prefix}", f"-DVVHIP_PREBUILT={vvhip}",
                    f"-DCMAKE_INSTALL_PREFIX={os.path.join(work, 'install')}"], check=True)
    subprocess.run(["cmake", "--build", os.path.join(work, "build"), "-j", "4"], check=True)
    py = os.path.join(work, "python")
    os.makedirs(py, exist_ok=True)
    for f in ("velocityverletplugin.i", "setup.py"):
        shutil.copy(os.path.join(ROOT, "python", f), py)
    subprocess.run(["swig", "-python", "-c++", f"-I{prefix}/include", f"-I{ROOT}/openmmapi/include", "-o", "VVPluginWrapper.cpp", "velocityverletplugin.i"],
                   cwd=py, check=True)
    libdir = os.path.join(work, "lib")                       # setup.py looks for libOpenMMVelocityVerlet in $VV_DIR/lib
    os.makedirs(libdir, exist_ok=True)
    shutil.copy(os.path.join(work, "build", "libOpenMMVelocityVerlet.so"), libdir)
    shutil.copy(vvhip, libdir)
    shutil.copytree(os.path.join(ROOT, "openmmapi"), os.path.join(work, "openmmapi"), dirs_exist_ok=True)
    env = dict(os.environ, OPENMM_DIR=prefix, VV_DIR=work)
    subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=py, env=env, check=True)
    return py, os.path.join(work, "build", "plugins")


def test_run_bulk_style_script_steps_the_c3_box_on_the_hip_platform(tmp_path):
    pre, why = _prerequisites()
    if pre is None:
        pytest.skip(f"real-OpenMM drop-in check not possible here: {why}")
    mm, prefix = pre
    from openmm import unit as u
    py, plugins = _build(prefix, str(tmp_path))
    sys.path.insert(0, py)
    mm.Platform.loadPluginsFromDirectory(plugins)            # what OpenMM does with <prefix>/lib/plugins at start-up
    from velocityverletplugin import VVIntegrator            # the SWIG class, as in run-bulk.py:66

    pkg = importlib.import_module("openmm-velocityverlet_amd")
    from oracle.make_golden_refhost import FULL, FULL_STEPS, inputs_for
    spec, params = FULL["C3_full"]()
    _, force = inputs_for(spec, params, FULL_STEPS)

    system = mm.System()
    for m in spec.masses:
        system.addParticle(float(m))
    lx, ly, lz = (float(b) for b in spec.box)
    system.setDefaultPeriodicBoxVectors(mm.Vec3(lx, 0, 0), mm.Vec3(0, ly, 0), mm.Vec3(0, 0, lz))
    drude = mm.DrudeForce()
    for d, par in np.asarray(spec.drude_pairs).reshape(-1, 2):
        drude.addParticle(int(d), int(par), -1, -1, -1, 0.0, 1e-3, 0.0, 0.0)      # charge 0: the pair list is what the integrator needs
    system.addForce(drude)
    # the recorded run has static forces: a per-particle constant force reproduces them (int64 / 2^32, planar)
    P = len(force) // 3
    fx, fy, fz = (force[k * P:k * P + spec.num_atoms].astype(np.float64) / 4294967296.0 for k in range(3))
    ext = mm.CustomExternalForce("-(fx*x+fy*y+fz*z)")
    for name in ("fx", "fy", "fz"):
        ext.addPerParticleParameter(name)
    for i in range(spec.num_atoms):
        ext.addParticle(i, [float(fx[i]), float(fy[i]), float(fz[i])])
    system.addForce(ext)
    # molecules: the reference takes them from the bonded topology (Context::getMolecules); zero-strength bonds chain each molecule's atoms
    bonds = mm.HarmonicBondForce()
    mol = np.asarray(spec.mol_id)
    for i in range(1, spec.num_atoms):
        if mol[i] == mol[i - 1]:
            bonds.addBond(i - 1, i, 0.1, 0.0)
    system.addForce(bonds)

    integrator = VVIntegrator(333.0 * u.kelvin, 10 / u.picosecond, 1 * u.kelvin, 40 / u.picosecond, 0.001 * u.picosecond)
    integrator.setUseMiddleScheme(True)
    integrator.setMaxDrudeDistance(0.02 * u.nanometer)
    platform = mm.Platform.getPlatformByName("HIP")
    context = mm.Context(system, integrator, platform, {"HipPrecision": "mixed"})
    context.setPositions(np.asarray(spec.positions, dtype=np.float64))
    context.setVelocities(np.asarray(spec.velocities, dtype=np.float64))
    integrator.step(FULL_STEPS)
    state = context.getState(getPositions=True, getVelocities=True)
    x = state.getPositions(asNumpy=True).value_in_unit(u.nanometer)
    v = state.getVelocities(asNumpy=True).value_in_unit(u.nanometer / u.picosecond)

    g = np.load(os.path.join(GOLDEN, "refhost_C3_full.npz"))
    idx = g["index"]
    vref = g["velm"][:, :3]
    xref = g["posq"][:, :3].astype(np.float64) + g["posq_corr"][:, :3].astype(np.float64)
    ev = np.abs(v[idx] - vref).max() / np.abs(vref).max()
    ex = np.abs(x[idx] - xref).max() / np.abs(xref).max()
    # north_star's bar; the product behind its own host agrees with this fixture to 5e-15 / 8e-16 (tests/test_gpu_ref_host.py) --
    # what is new here is OpenMM's own context, force evaluation and state transfer around the same kernels
    assert ev < 1e-5 and ex < 1e-5, (ev, ex)
    # the getters carry units, as the reference's SWIG module returns them (python/velocityverletplugin.i:35-79)
    assert integrator.getTemperature().unit == u.kelvin and abs(integrator.getMaxDrudeDistance().value_in_unit(u.nanometer) - 0.02) < 1e-12
