// TESTS ONLY — see opencv.hpp in this directory (declaration stubs; pins nothing).
#pragma once
#include "opencv.hpp"
