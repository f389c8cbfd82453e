#pragma once
#include "../../OpenMMCompat.h"
#ifndef OPENMM_EXPORT_DRUDE
#define OPENMM_EXPORT_DRUDE
#endif
#ifndef OPENMM_EXPORT
#define OPENMM_EXPORT
#endif
