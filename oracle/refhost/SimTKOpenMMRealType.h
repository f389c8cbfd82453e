// oracle/refhost stand-in for OpenMM's header of the same name (the reference includes it without the openmm/reference/ prefix)
#pragma once
#include "openmm/reference/SimTKOpenMMRealType.h"
