// stands where the reference's app/LinearHelperCKKS.cuh stands
#pragma once
#include "../src/troy_cuda.cuh"
#include "../../troyn_app.hpp"
