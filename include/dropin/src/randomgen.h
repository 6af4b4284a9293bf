// test/evaluator_cuda.cu:2460 includes ../src/randomgen.h and uses nothing from it: the CPU-side PRNG classes are not part of the device interface
#pragma once
