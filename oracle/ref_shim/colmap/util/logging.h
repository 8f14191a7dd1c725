#pragma once
#include "ref_shim_types.h"
