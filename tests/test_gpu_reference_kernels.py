"""-m gpu: the REFERENCE's own kernels (platforms/cuda/src/kernels/*.cu compiled unmodified for gfx950 by `make -C oracle
refgpu`, launched in the reference's order by oracle/ref_gpu_driver.cpp) running on the same GPU as the product.
Three-way check on identical seeded inputs: reference-kernels-on-GPU == oracle (CPU) == product, to 1e-5 relative
(measured ~1e-14: only reduction order differs)."""
import importlib

import numpy as np
import pytest

from oracle import oracle as O

pkg = importlib.import_module("openmm-velocityverlet_amd")
I, systems = pkg.integrator, pkg.systems
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref_gpu(), reason="oracle/_ref GPU build absent (needs /root/reference at build time)")]


@pytest.mark.parametrize("case,cos", [("bulk", 0.0), ("bulk", 0.02), ("water", 0.0), ("C3-full-size", 0.0), ("C3-full-size", 0.02)])
def test_reference_kernels_on_gpu_vs_oracle_vs_product(case, cos):
    if case == "C3-full-size":              # BASELINE.json's headline box: 111 000 particles through the reference's own kernels
        spec, T, maxd = systems.make_config("C3"), 333.0, 0.02
    elif case == "bulk":
        spec, T, maxd = systems.drude_il(cells=(1, 1, 1), pairs_per_cell=60, seed=41), 333.0, 0.02
    else:
        spec, T, maxd = systems.spce_water(400, seed=42), 300.0, 0.0
    nsteps = 20
    p = O.Params(temperature=T, drude_temperature=1.0, max_drude_distance=maxd, cos_acceleration=cos)
    ref = O.RefGpuSystem(spec, p)
    ref.step(nsteps)
    r = ref.download()
    ref.close()
    osys = O.OracleSystem(spec, p, "mixed", force_mode=1)
    osys.step(nsteps)
    it = I.VVIntegrator(T, 10.0, 1.0, 40.0, 0.001)
    it.setMaxDrudeDistance(maxd)
    it.setCosAcceleration(cos)
    ctx = I.Context(spec, it, precision="mixed")
    it.step(nsteps)
    x_p, v_p = ctx.getPositions(), ctx.getVelocities()
    ctx.close()
    def rel(a, b):
        return np.abs(a - b).max() / np.abs(b).max()
    e_ro = max(rel(r["positions"], osys.positions()), rel(r["velm"][:, :3], osys.velm[:, :3]))
    e_pr = max(rel(x_p, r["positions"]), rel(v_p, r["velm"][:, :3]))
    print(f"{case} cos={cos}: reference-on-GPU vs oracle {e_ro:.1e}; product vs reference-on-GPU {e_pr:.1e}")
    assert e_ro < 1e-5 and e_pr < 1e-5
    assert np.allclose(r["ke2"][:osys.s.num_tg], osys.ke2()[:osys.s.num_tg], rtol=1e-9)
