#pragma once
#include "../../OpenMMCompat.h"
#ifndef OPENMM_EXPORT_DRUDE
#define OPENMM_EXPORT_DRUDE
#endif
#ifndef OPENMM_EXPORT
#define OPENMM_EXPORT
#endif
// OpenMM's index check as the reference uses it (openmmapi/src/VVIntegrator.cpp:83,88): throws OpenMMException when out of range
#ifndef ASSERT_VALID_INDEX
#define ASSERT_VALID_INDEX(index, vector) { if ((index) < 0 || (index) >= (long long) (vector).size()) throw OpenMM::OpenMMException("Assertion failure: Index out of range"); }
#endif
