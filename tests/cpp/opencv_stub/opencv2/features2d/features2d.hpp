// see ../core/core.hpp: signature-only stand-in, test infrastructure
#pragma once
#include "../core/core.hpp"
