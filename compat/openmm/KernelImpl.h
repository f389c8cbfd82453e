#pragma once
#include "../OpenMMCompat.h"
