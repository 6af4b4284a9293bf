// gtest_main of the shim (tests/cpp/gtest_shim/gtest/gtest.h): walks the TEST registry.  TEST INFRASTRUCTURE ONLY.
#include "gtest/gtest.h"
int main(int argc, char **argv) { return ::gtest_shim::run_all(argc, argv); }
