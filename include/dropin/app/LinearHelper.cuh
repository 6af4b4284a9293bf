// stands where the reference's app/LinearHelper.cuh stands
#pragma once
#include "../src/troy_cuda.cuh"
#include "../../troyn_linear.hpp"
