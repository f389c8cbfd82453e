"""Host-side mirror of the reference's Python surface for the hot path.

``VVIntegrator`` has the methods of the SWIG class ``velocityverletplugin.VVIntegrator``
(/root/reference/python/velocityverletplugin.i:83-129) with the argument meaning and defaults of the
C++ class (openmmapi/include/openmm/VVIntegrator.h:49-507, openmmapi/src/VVIntegrator.cpp:46-70).
Because neither OpenMM nor its unit package exists here, getters return plain floats in OpenMM's MD
units (K, 1/ps, nm, nm/ps^2, kJ/(nm e)) instead of ``unit.Quantity``.

``Context`` stands in for ``openmm.Context`` just far enough to run the path stand-alone: it owns the
device arrays OpenMM's HipContext would own (velm, posq, posqCorrection, force, random) and a force
provider that plays the part of ``calcForcesAndEnergy``.  Inside a real OpenMM the same C ABI is driven
by the C++ adapters in platforms/hip.  All arithmetic happens in libvvhip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from . import vvhip as H
from .systems import SystemSpec


def padded(n: int) -> int:
    return (n + 31) // 32 * 32


class VVIntegrator:
    """Same constructor and method names as the reference's VVIntegrator (velocityverletplugin.i:85-127)."""

    def __init__(self, temperature, frequency, drudeTemperature, drudeFrequency, stepSize, numNHChains=3, loopsPerStep=1):
        # openmmapi/src/VVIntegrator.cpp:46-70
        self._temperature = float(temperature)
        self._frequency = float(frequency)
        self._drudeTemperature = float(drudeTemperature)
        self._drudeFrequency = float(drudeFrequency)
        self._stepSize = float(stepSize)
        self._numNHChains = int(numNHChains)
        self._loopsPerStep = int(loopsPerStep)
        self._constraintTolerance = 1e-5
        self._maxDrudeDistance = 0.0
        self._friction = 5.0
        self._drudeFriction = 20.0
        self._randomNumberSeed = 0
        self._mirrorLocation = 0.0
        self._electricField = 0.0
        self._cosAcceleration = 0.0
        self._useCOMTempGroup = False
        self._useMiddleScheme = True
        self._debugEnabled = False
        self._autoSetCOMTempGroup = True
        self._autoSetFriction = True
        self._particlesLD: List[int] = []
        self._imagePairs: List[Tuple[int, int]] = []
        self._particlesElectrolyte: List[int] = []
        self._context: Optional["Context"] = None

    # ---- plain parameters (VVIntegrator.h:70-431)
    def getTemperature(self): return self._temperature
    def setTemperature(self, temp): self._temperature = float(temp); self._push()
    def getFrequency(self): return self._frequency
    def setFrequency(self, tau): self._frequency = float(tau); self._push()
    def getDrudeTemperature(self): return self._drudeTemperature
    def setDrudeTemperature(self, temp): self._drudeTemperature = float(temp); self._push()
    def getDrudeFrequency(self): return self._drudeFrequency
    def setDrudeFrequency(self, tau): self._drudeFrequency = float(tau); self._push()
    def getStepSize(self): return self._stepSize
    def setStepSize(self, dt): self._stepSize = float(dt); self._push()
    def getConstraintTolerance(self): return self._constraintTolerance
    def setConstraintTolerance(self, tol): self._constraintTolerance = float(tol)
    def getNumNHChains(self): return self._numNHChains
    def setNumNHChains(self, n): self._numNHChains = int(n)
    def getLoopsPerStep(self): return self._loopsPerStep
    def setLoopsPerStep(self, n): self._loopsPerStep = int(n); self._push()
    def getUseCOMTempGroup(self): return self._useCOMTempGroup

    def setUseCOMTempGroup(self, use):                      # VVIntegrator.h:147-150
        self._useCOMTempGroup = bool(use)
        self._autoSetCOMTempGroup = False

    def getUseMiddleScheme(self): return self._useMiddleScheme
    def setUseMiddleScheme(self, use): self._useMiddleScheme = bool(use)
    def getMaxDrudeDistance(self): return self._maxDrudeDistance
    def setMaxDrudeDistance(self, d): self._maxDrudeDistance = float(d); self._push()

    def addParticleLangevin(self, particle):                # VVIntegrator.h:186-189
        self._particlesLD.append(int(particle))
        return len(self._particlesLD)

    def getRandomNumberSeed(self): return self._randomNumberSeed
    def setRandomNumberSeed(self, seed): self._randomNumberSeed = int(seed)
    def getFriction(self): return self._friction

    def setFriction(self, fric):                            # VVIntegrator.h:213-216 (quirk Q7: double, not int)
        self._friction = float(fric)
        self._autoSetFriction = False
        self._push()

    def getDrudeFriction(self): return self._drudeFriction

    def setDrudeFriction(self, fric):
        self._drudeFriction = float(fric)
        self._autoSetFriction = False
        self._push()

    def addImagePair(self, image, parent):                  # VVIntegrator.cpp:76-80
        self._imagePairs.append((int(image), int(parent)))
        return len(self._imagePairs)

    def getImagePairs(self): return list(self._imagePairs)
    def setMirrorLocation(self, z): self._mirrorLocation = float(z); self._push()
    def getMirrorLocation(self): return self._mirrorLocation

    def addParticleElectrolyte(self, particle):
        self._particlesElectrolyte.append(int(particle))
        return len(self._particlesElectrolyte)

    def setElectricField(self, field):
        """kJ/(nm e) per particle, as the C++ API (1 V/nm = 1.602176634e-22 here; quirk Q12)."""
        self._electricField = float(field)
        self._push()

    def getElectricField(self): return self._electricField
    def setCosAcceleration(self, a): self._cosAcceleration = float(a); self._push()
    def getCosAcceleration(self): return self._cosAcceleration
    def getDebugEnabled(self): return self._debugEnabled
    def setDebugEnabled(self, e): self._debugEnabled = bool(e)

    # ---- actions
    def step(self, steps):                                  # VVIntegrator.cpp:223-230
        if self._context is None:
            raise H.VVHipError(H.ERR_INVALID, "This Integrator is not bound to a context!")
        self._context._step(int(steps))

    def getViscosity(self):                                 # VVIntegrator.cpp:378-383 -> (vMax nm/ps, 1/viscosity)
        if self._context is None or self._cosAcceleration == 0:
            return (0.0, 0.0)
        return self._context._viscosity()

    # ---- internals
    def _params(self) -> H.Params:
        return H.Params(self._temperature, self._frequency, self._drudeTemperature, self._drudeFrequency, self._stepSize,
                        self._numNHChains, self._loopsPerStep, self._maxDrudeDistance, self._friction, self._drudeFriction,
                        self._mirrorLocation, self._electricField, self._cosAcceleration, int(self._useCOMTempGroup),
                        int(self._useMiddleScheme), int(self._autoSetCOMTempGroup), int(self._autoSetFriction), self._constraintTolerance)

    def _push(self):
        if self._context is not None:
            self._context._set_params()


def create_plan(system: SystemSpec, integrator: "VVIntegrator", precision: str = "mixed", shard=None,
                particles_ld=None, image_pairs=None, electrolyte=None):
    """vvhip_plan_create: host-only analysis (no GPU needed).  Returns (plan handle, PlanInfo, keep-alive arrays)."""
    n = system.num_atoms
    shard = (0, n) if shard is None else shard
    particles_ld = list(system.particles_ld) + list(integrator._particlesLD) if particles_ld is None else particles_ld
    image_pairs = list(system.image_pairs) + list(integrator._imagePairs) if image_pairs is None else image_pairs
    electrolyte = list(system.particles_electrolyte) + list(integrator._particlesElectrolyte) if electrolyte is None else electrolyte
    k = dict(
        masses=np.ascontiguousarray(system.masses, dtype=np.float64),
        mol_id=np.ascontiguousarray(system.mol_id, dtype=np.int32),
        drude=np.ascontiguousarray(system.drude_pairs, dtype=np.int32).reshape(-1),
        cons=np.ascontiguousarray(system.constraints, dtype=np.int32).reshape(-1),
        ld=np.ascontiguousarray(particles_ld, dtype=np.int32),
        img=np.ascontiguousarray(image_pairs, dtype=np.int32).reshape(-1),
        el=np.ascontiguousarray(electrolyte, dtype=np.int32),
        cdist=np.ascontiguousarray(getattr(system, "constraint_distances", None) if getattr(system, "constraint_distances", None) is not None else [], dtype=np.float64))
    if k["cdist"].size not in (0, k["cons"].size // 2):
        raise ValueError("constraint_distances must have one entry per constraint")
    vsites = list(getattr(system, "virtual_sites", None) or [])      # (site, kind, (parents), (params)) per site: System::getVirtualSite
    k["vs"] = np.full((len(vsites), 5), -1, dtype=np.int32)
    k["vsp"] = np.zeros((len(vsites), 12), dtype=np.float64)
    for i, (site, kind, parents, prm) in enumerate(vsites):
        k["vs"][i, 0], k["vs"][i, 1] = site, kind
        k["vs"][i, 2:2 + len(parents)] = parents
        k["vsp"][i, :len(prm)] = prm
    ptr = lambda a: a.ctypes.data if a.size else None
    desc = H.SystemDesc(n, padded(n), ptr(k["masses"]), ptr(k["mol_id"]), system.num_molecules,
                        k["drude"].size // 2, ptr(k["drude"]), k["cons"].size // 2, ptr(k["cons"]),
                        int(system.has_cm_motion_remover), k["ld"].size, ptr(k["ld"]), k["img"].size // 2, ptr(k["img"]),
                        k["el"].size, ptr(k["el"]), int(shard[0]), int(shard[1]), ptr(k["cdist"]),
                        len(vsites), ptr(k["vs"]), ptr(k["vsp"]))
    plan = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = H.lib.vvhip_plan_create(C.byref(desc), C.byref(integrator._params()), H.PRECISION[precision], C.byref(plan), err, 512)
    if rc != H.OK:
        raise H.VVHipError(rc, err.value.decode())
    info = H.PlanInfo()
    H.check(H.lib.vvhip_plan_get_info(plan, C.byref(info)), plan)
    return plan, info, k


def plan_layout(system: SystemSpec, integrator: "VVIntegrator", precision: str = "mixed", shard=None):
    """Host-only: the analysis results and the wave layout [num_waves*64, 2] = (particle, role word)."""
    plan, info, _ = create_plan(system, integrator, precision, shard)
    try:
        nslots = H.lib.vvhip_plan_get_slots(plan, None, 0)
        slots = np.zeros((nslots, 2), dtype=np.int32)
        got = H.lib.vvhip_plan_get_slots(plan, slots.ctypes.data, nslots)
        assert got == nslots
    finally:
        H.lib.vvhip_plan_destroy(plan)
    return info, slots


def plan_launch_shape(system: SystemSpec, integrator: "VVIntegrator", precision: str = "mixed", shard=None):
    """Host-only: (threads of a block's tile waves, most blocks of kernel A, of kernel B, tile waves per block of the one-launch step or 0) as the
    plan chooses them for a whole MI355X (vvhip_debug_launch_shape)."""
    plan, _, _ = create_plan(system, integrator, precision, shard)
    try:
        shape = (C.c_int32 * 4)()
        H.check(H.lib.vvhip_debug_launch_shape(plan, C.byref(shape)), plan)
    finally:
        H.lib.vvhip_plan_destroy(plan)
    return tuple(shape)


DEFAULT_TUNE: dict = {}      # see Context(tune=...)


class Context:
    """Device state + force provider around one VVIntegrator.

    force_provider: "tether" (synthetic forces recomputed from the positions before every force
    evaluation, see vvhip_synth_tether_force), "static" (whatever is in the force buffer), or a callable
    ``f(context)`` that fills ``context.force`` on the device.
    """

    def __init__(self, system: SystemSpec, integrator: VVIntegrator, precision: str = "mixed",
                 force_provider="tether", k_tether: float = 1000.0, k_drude: float = 209200.0,
                 random: Optional[np.ndarray] = None, shard: Optional[Tuple[int, int]] = None, device: Optional[int] = None,
                 stream: Optional[int] = None, tune: Optional[dict] = None):
        """tune: {name: value} for vvhip_debug_tune (test hook: launch shape, loaded instead of computed slot words, ...), applied between
        plan creation and binding, on top of the module's DEFAULT_TUNE (which tests patch to reach contexts created elsewhere)."""
        if integrator._context is not None:
            raise H.VVHipError(H.ERR_INVALID, "This Integrator is already bound to a context")   # VVIntegrator.cpp:93-94
        if H.device_count() == 0:
            raise H.VVHipError(H.ERR_NO_DEVICE, "no HIP device visible: the hot path has no CPU fallback")
        if device is not None:
            H.check(H.lib.vvhip_set_device(int(device)), what="hipSetDevice failed")
        self.system, self.integrator, self.precision = system, integrator, precision
        self.force_provider, self.k_tether, self.k_drude = force_provider, float(k_tether), float(k_drude)
        # particles added through the integrator API extend what the system spec lists
        self._particles_ld = list(system.particles_ld) + list(integrator._particlesLD)
        self._image_pairs = list(system.image_pairs) + list(integrator._imagePairs)
        self._electrolyte = list(system.particles_electrolyte) + list(integrator._particlesElectrolyte)
        n = system.num_atoms
        self.shard = (0, n) if shard is None else (int(shard[0]), int(shard[1]))
        self.plan, self.info, self._keep = create_plan(system, integrator, precision, self.shard, self._particles_ld,
                                                       self._image_pairs, self._electrolyte)
        plan = self.plan
        if getattr(system, "virtual_sites", None) and self.info.num_virtual_sites == 0:
            # (in the OpenMM plugin the adapters then keep calling OpenMM's computeVirtualSites; this stand-alone host has no such kernel)
            H.lib.vvhip_plan_destroy(plan)
            self.plan = None
            raise H.VVHipError(H.ERR_UNSUPPORTED,
                               "the System's virtual sites cannot be placed by kernel B (a site on a site, or parents outside the site's wave) "
                               "and this host has no computeVirtualSites of its own")
        for key, value in {**DEFAULT_TUNE, **(tune or {})}.items():
            H.check(H.lib.vvhip_debug_tune(plan, key.encode(), int(value)), plan)

        # ---- device arrays in OpenMM's layouts, shard-local
        R, M = H.REAL[precision], H.MIXED_T[precision]
        b, e = self.shard
        nloc = e - b
        self.nloc = nloc
        velm = np.zeros((nloc, 4), dtype=M)
        velm[:, :3] = system.velocities[b:e]
        m = system.masses[b:e]
        velm[:, 3] = np.where(m != 0, 1.0 / np.where(m != 0, m, 1.0), 0.0)
        posq = np.zeros((nloc, 4), dtype=R)
        posq[:, :3] = system.positions[b:e]
        posq[:, 3] = system.charges[b:e]
        corr = np.zeros((nloc, 4), dtype=R)
        if precision == "mixed":
            corr[:, :3] = system.positions[b:e] - posq[:, :3].astype(np.float64)
        self.padded = padded(n)
        self.velm = H.DeviceArray.from_host(velm)
        self.posq = H.DeviceArray.from_host(posq)
        self.posq_corr = H.DeviceArray.from_host(corr)
        self.site = H.DeviceArray.from_host(posq)
        self.force = H.DeviceArray.from_host(np.zeros(3 * self.padded, dtype=np.int64))
        self.pos_delta = H.DeviceArray.from_host(np.zeros((nloc, 4), dtype=M))
        self._random_injected = random is not None
        if random is None and self.info.num_normal_ld + self.info.num_pairs_ld > 0:
            random = np.random.default_rng(1).standard_normal((1 << 16, 4)).astype(np.float32)   # seed 1
        self.random_host = None if random is None else np.ascontiguousarray(random, dtype=np.float32)
        self.random = None if random is None else H.DeviceArray.from_host(self.random_host)
        self.random_index = 0
        self._own_stream = None
        if stream is None:
            s = C.c_void_p()
            H.check(H.lib.vvhip_stream_create(C.byref(s)), what="hipStreamCreate failed")
            self._own_stream = s.value
            stream = s.value
        self.stream = stream
        buf = H.Buffers(self.velm.ptr, self.posq.ptr, self.posq_corr.ptr if precision == "mixed" else None, self.force.ptr,
                        self.pos_delta.ptr, self.random.ptr if self.random is not None else None,
                        0 if self.random is None else self.random_host.shape[0], stream)
        H.check(H.lib.vvhip_bind(plan, C.byref(buf)), plan)
        box = (C.c_double * 3)(*[float(x) for x in system.box])
        H.check(H.lib.vvhip_set_box(plan, C.byref(box)), plan)
        self.forces_valid = False
        integrator._context = self

    # ---- state access (blocking)
    def synchronize(self):
        H.check(H.lib.vvhip_synchronize(self.plan), self.plan)

    def getVelm(self): self.synchronize(); return self.velm.download()
    def getPosq(self): self.synchronize(); return self.posq.download()
    def getPosqCorrection(self): self.synchronize(); return self.posq_corr.download()
    def getForce(self): self.synchronize(); return self.force.download()

    def getPositions(self):
        p = self.getPosq()[:, :3].astype(np.float64)
        if self.precision == "mixed":
            p = p + self.getPosqCorrection()[:, :3].astype(np.float64)
        return p

    def getVelocities(self): return self.getVelm()[:, :3].astype(np.float64)

    def setVelocities(self, v):
        velm = self.getVelm()
        velm[:, :3] = v
        self.velm.upload(velm)
        self.forces_valid = False                               # VVIntegrator.h:447-449 stateChanged

    def getKineticEnergy(self) -> float:
        """1/2 sum m v^2 [kJ/mol] (what State.getKineticEnergy() returns through VVIntegrator::computeKineticEnergy)."""
        ke = C.c_double()
        H.check(H.lib.vvhip_compute_kinetic_energy(self.plan, C.byref(ke)), self.plan)
        return ke.value

    def getGroupTemperatures(self):
        """Temperatures of the thermostat groups [atom, COM, Drude] at the last thermostat application: 2KE_g / (dof_g kB) --
        the quantities examples/ommhelper/reporter/drudetemperaturereporter.py:98-133 recomputes on the host in NumPy."""
        st = self.getNHState()
        kb = 8.31446261815324e-3
        return [st.ke2[g] / (self.info.dof[g] * kb) if self.info.dof[g] > 0 else 0.0 for g in range(3)]

    def getNHState(self) -> H.NHState:
        s = H.NHState()
        H.check(H.lib.vvhip_get_nh_state(self.plan, C.byref(s)), self.plan)
        return s

    def setNHState(self, s: H.NHState):
        H.check(H.lib.vvhip_set_nh_state(self.plan, C.byref(s)), self.plan)

    # ---- stepping
    def setPeriodicBoxSize(self, lx, ly, lz):
        """A barostat move as the plugin sees it: cu.getPeriodicBoxSize() is read live by the cos kernels (HOST:1057, 1129)."""
        box = (C.c_double * 3)(float(lx), float(ly), float(lz))
        H.check(H.lib.vvhip_set_box(self.plan, C.byref(box)), self.plan)

    def _set_params(self):
        H.check(H.lib.vvhip_set_params(self.plan, C.byref(self.integrator._params())), self.plan)

    def calcForces(self):
        """Plays context->calcForcesAndEnergy (OpenMM's, out of scope for the plugin)."""
        if self.force_provider == "tether":
            H.check(H.lib.vvhip_synth_tether_force(self.plan, self.site.ptr, self.k_tether, self.k_drude), self.plan)
        elif callable(self.force_provider):
            self.force_provider(self)

    def _prepare_random(self) -> int:
        """integration.prepareRandomNumbers(n) of OpenMM, on a fixed buffer (HOST:863)."""
        if self.random is None:
            return 0
        cnt = max(self.info.num_normal_ld, 1) + 2 * max(self.info.num_pairs_ld, 1)
        if self.random_index + cnt <= self.random_host.shape[0]:
            old = self.random_index
            self.random_index += cnt
            return old
        # exhausted: OpenMM's prepareRandomNumbers refills the buffer here; so does this host, with the device generator -- unless the
        # caller injected the normals (parity runs against the oracle, which has no generator and starts over at index 0)
        if not self._random_injected:
            self.fill_random()
        self.random_index = cnt
        return 0

    def _step(self, steps: int):
        L, plan = H.lib, self.plan
        has_ld = self.info.num_normal_ld + self.info.num_pairs_ld > 0
        for _ in range(steps):
            if self.integrator._useMiddleScheme:               # VVIntegrator.cpp:232-270
                self.calcForces()
                ri = self._prepare_random() if has_ld else 0
                H.check(L.vvhip_step_middle(plan, ri), plan)
            else:                                              # VVIntegrator.cpp:272-338
                if not self.forces_valid:
                    self.calcForces()
                    self.forces_valid = True
                H.check(L.vvhip_step_vv_first(plan), plan)
                self.calcForces()
                self.forces_valid = True
                ri = self._prepare_random() if has_ld else 0
                H.check(L.vvhip_step_vv_second(plan, ri), plan)

    def run_graph(self, steps: int, steps_per_graph: int = 50):
        """Replay a captured hipGraph of whole steps (force provider included; both schemes).  With a Langevin subset the
        graph starts with a refill of the random buffer by the device generator, so every replay draws fresh numbers."""
        site = self.site.ptr if self.force_provider == "tether" else None
        H.check(H.lib.vvhip_run_graph(self.plan, int(steps), int(steps_per_graph), site, self.k_tether, self.k_drude), self.plan)
        if not self.integrator._useMiddleScheme and site is not None and steps > 0:
            self.forces_valid = True                                # the last force evaluation was at the final positions

    def graph_prepare(self, steps_per_graph: int = 50):
        """Capture + instantiate + upload the graph for the current thermostat parity without launching it (vvhip_graph_prepare):
        hosts call it after their warm-up so that no capture ever falls into a timed region."""
        site = self.site.ptr if self.force_provider == "tether" else None
        H.check(H.lib.vvhip_graph_prepare(self.plan, int(steps_per_graph), site, self.k_tether, self.k_drude), self.plan)

    def status(self):
        """(mailbox_timed_out, accumulator_overflow): the sticky health flags, read without synchronising."""
        a, b = C.c_int32(0), C.c_int32(0)
        H.check(H.lib.vvhip_status(self.plan, C.byref(a), C.byref(b)), self.plan)
        return bool(a.value), bool(b.value)

    def status_clear(self):
        H.check(H.lib.vvhip_status_clear(self.plan), self.plan)

    def status_words(self):
        """[mailbox wait ran out, accumulator overflow, one-launch rendezvous ran out, constraint cluster hit its iteration cap]: all four
        sticky health words (vvhip_status_words)."""
        w = (C.c_int32 * 4)()
        H.check(H.lib.vvhip_status_words(self.plan, C.byref(w)), self.plan)
        return [int(x) for x in w]

    def fused_status(self):
        """(active, launches): whether vvhip_step_middle runs this plan's step as ONE launch (kernels A and B around an in-kernel
        rendezvous), and how many such launches it has enqueued (vvhip_fused_status)."""
        a, n = C.c_int32(0), C.c_int64(0)
        H.check(H.lib.vvhip_fused_status(self.plan, C.byref(a), C.byref(n), None), self.plan)
        return bool(a.value), int(n.value)

    def recovery_count(self) -> int:
        """How often this plan has repaired a missed rendezvous of the one-launch step (include/vvhip.h: vvhip_recovery_count)."""
        n = C.c_int64(0)
        H.check(H.lib.vvhip_recovery_count(self.plan, C.byref(n)), self.plan)
        return int(n.value)

    def fused_wait_units(self) -> int:
        """Where the self-tuning wait of the one-launch step's rendezvous stands (units of 256 shader clocks; synchronises)."""
        w = C.c_int32(0)
        H.check(H.lib.vvhip_fused_status(self.plan, None, None, C.byref(w)), self.plan)
        return int(w.value)

    def fill_random(self, seed=None):
        """Refill the Langevin random buffer with the device generator (Philox4x32-10 + Box-Muller)."""
        if seed is not None:
            H.check(H.lib.vvhip_set_random_seed(self.plan, int(seed)), self.plan)
        H.check(H.lib.vvhip_fill_random(self.plan), self.plan)

    def run_eager(self, steps: int):
        """The same steps enqueued one by one from C (no per-step Python, no graph; both schemes)."""
        site = self.site.ptr if self.force_provider == "tether" else None
        H.check(H.lib.vvhip_run_eager(self.plan, int(steps), site, self.k_tether, self.k_drude), self.plan)
        if not self.integrator._useMiddleScheme and site is not None and steps > 0:
            self.forces_valid = True

    def run_eager_unfused(self, steps: int):
        """Middle scheme through the per-KernelImpl entry points in VVIntegrator::stepMiddle's order, enqueued from C: the launches
        of the un-fused drop-in path (a host solver between the stages would add its own)."""
        site = self.site.ptr if self.force_provider == "tether" else None
        H.check(H.lib.vvhip_run_eager_unfused(self.plan, int(steps), site, self.k_tether, self.k_drude), self.plan)

    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        """Give the plan an RCCL communicator (vvhip_comm_init); the id comes from comm_unique_id() on rank 0."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        H.check(H.lib.vvhip_comm_init(self.plan, buf, int(nranks), int(rank)), self.plan)

    def comm_count(self) -> int:
        """Ranks of the plan's RCCL communicator as ncclCommCount reports them (0: none)."""
        n = C.c_int32(0)
        H.check(H.lib.vvhip_comm_count(self.plan, C.byref(n)), self.plan)
        return n.value

    def mailbox_create(self, nranks: int, rank: int) -> bytes:
        """xGMI mailbox exchange (vvhip_mailbox_*): returns this rank's 64-byte IPC handle; gather all ranks' handles in rank
        order and pass them to mailbox_connect()."""
        buf = C.create_string_buffer(64)
        H.check(H.lib.vvhip_mailbox_create(self.plan, int(nranks), int(rank), buf), self.plan)
        return buf.raw

    def mailbox_connect(self, handles: bytes):
        buf = C.create_string_buffer(bytes(handles), len(handles))
        H.check(H.lib.vvhip_mailbox_connect(self.plan, buf), self.plan)

    def mailbox_status(self):
        """(active, timed_out): whether the fused steps use the mailbox, and whether a wait on the peers ever ran out."""
        a, t = C.c_int32(0), C.c_int32(0)
        H.check(H.lib.vvhip_mailbox_status(self.plan, C.byref(a), C.byref(t)), self.plan)
        return bool(a.value), bool(t.value)

    def mailbox_layout(self):
        """(shared_device, arithmetic_layout): whether a peer's box lives on this rank's own device, and whether kernel B takes the
        arithmetic work-item layout next to the exchange."""
        a, t = C.c_int32(0), C.c_int32(0)
        H.check(H.lib.vvhip_mailbox_layout(self.plan, C.byref(a), C.byref(t)), self.plan)
        return bool(a.value), bool(t.value)

    def mailbox_destroy(self):
        H.check(H.lib.vvhip_mailbox_destroy(self.plan), self.plan)

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        H.check(H.lib.vvhip_comm_unique_id(buf), what="ncclGetUniqueId failed (is librccl available?)")
        return buf.raw

    def _viscosity(self):
        v, inv = C.c_double(), C.c_double()
        H.check(H.lib.vvhip_calc_viscosity(self.plan, C.byref(v), C.byref(inv)), self.plan)
        return (v.value, inv.value)

    def generic_launches(self):
        """((count_A, count_B), (stage_set_A, stage_set_B)): launches of this plan that found no compiled specialisation of their stage
        set and ran the generic kernel (vvhip_generic_launches)."""
        n, f = (C.c_int64 * 2)(), (C.c_uint32 * 2)()
        H.check(H.lib.vvhip_generic_launches(self.plan, C.byref(n), C.byref(f)), self.plan)
        return (int(n[0]), int(n[1])), (int(f[0]), int(f[1]))

    @staticmethod
    def rtc_stats():
        """(kernels compiled at run time, launches of kernel A that ran one, of kernel B, seconds spent compiling), process-wide
        (vvhip_rtc_stats)."""
        n, t = (C.c_int64 * 3)(), C.c_double()
        rc = H.lib.vvhip_rtc_stats(C.byref(n), C.byref(t))
        if rc != 0:
            raise RuntimeError("vvhip_rtc_stats failed: %d" % rc)
        return int(n[0]), int(n[1]), int(n[2]), float(t.value)

    @staticmethod
    def rtc_mode(mode=-1):
        """Sets VVHIP_RTC's value for the launches that follow (0 never, 1 stage sets without a compiled kernel, 2 always) and returns
        the previous one; -1 only reads it (vvhip_rtc_mode)."""
        return int(H.lib.vvhip_rtc_mode(int(mode)))

    def timing(self, enable):
        """0 / False off; 1 / True every launch group; 2 kernels A and B only (dispatch timestamps, nothing added to the stream);
        n > 2 as 2 with n events prepared beforehand (vvhip_timing_enable)."""
        H.check(H.lib.vvhip_timing_enable(self.plan, int(enable)), self.plan)

    def timing_read(self):
        a, b, o = C.c_double(), C.c_double(), C.c_double()
        n = (C.c_int32 * 3)()
        H.check(H.lib.vvhip_timing_read(self.plan, C.byref(a), C.byref(b), C.byref(o), C.byref(n)), self.plan)
        return dict(ms_a=a.value, ms_b=b.value, ms_other=o.value, launches=list(n))

    def algorithmic_bytes(self):
        """(bytes_A, bytes_B) per particle of the fused middle step's two kernels (vvhip_algorithmic_bytes)."""
        a, b = C.c_int32(0), C.c_int32(0)
        H.check(H.lib.vvhip_algorithmic_bytes(self.plan, C.byref(a), C.byref(b)), self.plan)
        return a.value, b.value

    def time_kernel(self, kernel: int, reps: int = 200) -> float:
        """Average ms of `reps` back-to-back launches of stage kernel 0 (A) / 1 (B) with the fused step's stage bits.
        Timing only: the physical state is not meaningful afterwards."""
        ms = C.c_double()
        H.check(H.lib.vvhip_time_kernel(self.plan, int(kernel), 0xFFFFFFFF, int(reps), C.byref(ms)), self.plan)
        return ms.value

    def close(self):
        if getattr(self, "plan", None):
            H.lib.vvhip_comm_destroy(self.plan)
            H.lib.vvhip_plan_destroy(self.plan)
            self.plan = None
            self.integrator._context = None
            if self._own_stream:
                H.lib.vvhip_stream_destroy(self._own_stream)
                self._own_stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
