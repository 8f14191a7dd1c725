#pragma once
#include "glomap/mock_types.h"
