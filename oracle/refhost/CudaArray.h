// oracle/refhost stand-in: everything lives in CudaContext.h
#pragma once
#include "CudaContext.h"
