// compat stand-in for OpenMM's openmm/reference/SimTKOpenMMRealType.h: only the physical constants the reference's plugin uses
// (openmmapi/src/VVIntegrator.cpp:40,371; platforms/cuda/src/CudaVVKernels.cpp:40,190,583,837,978), with the CODATA-2018 values and
// the derivation order OpenMM 8.x uses (BOLTZ = BOLTZMANN * AVOGADRO / KILO).  Nothing here is OpenMM source.
#pragma once
#define ANGSTROM     (1e-10)
#define KILO         (1e3)
#define NANO         (1e-9)
#define PICO         (1e-12)
#define A2NM         (ANGSTROM/NANO)
#define NM2A         (NANO/ANGSTROM)
#define RAD2DEG      (180.0/M_PI)
#define CAL2JOULE    (4.184)
#define E_CHARGE     (1.602176634e-19)
#define AMU          (1.66053906660e-27)
#define BOLTZMANN    (1.380649e-23)            /* (J/K)   */
#define AVOGADRO     (6.02214076e23)
#define RGAS         (BOLTZMANN*AVOGADRO)      /* (J/(mol K))  */
#define BOLTZ        (RGAS/KILO)               /* (kJ/(mol K)) */
