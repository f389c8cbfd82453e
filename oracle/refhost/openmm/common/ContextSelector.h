// oracle/refhost stand-in: ContextSelector lives in CudaContext.h
#pragma once
#include "CudaContext.h"
