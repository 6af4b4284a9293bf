// stands where the reference's src/troy_cuda.cuh stands: callers that spell #include "../src/troy_cuda.cuh" get the MI355X library
#pragma once
#include "../../troy_cuda.cuh"
