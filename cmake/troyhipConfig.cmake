# troyhipConfig.cmake -- find_package(troyhip CONFIG) for users of the drop-in.
#
#   cmake -Dtroyhip_DIR=<this repo>/cmake ...          then          target_link_libraries(app PRIVATE troyhip::troyhip)
#
# troyhip::troyhip = libtroyhip.so (the gfx950 build made by `make -C troy_amd/csrc` / __graft_entry__.build()) + include/, where troy_cuda.cuh stands
# for the reference's src/troy_cuda.cuh (troyn:: classes, troy::*Cuda aliases), troyn.hpp / troyn_app.hpp / troyn_linear.hpp / troyn_devices.hpp are the
# C++ surface and troyhip.h is the C ABI.  The headers are plain C++17: user code is compiled by the HOST compiler (LANGUAGE CXX also for files named
# *.cu), only the library holds device code.
#   TROYHIP_LIBRARY   override the library file (e.g. tests/emul/libtroyhip_emul.so, the host emulator build of the test suite)
#   TROYHIP_DROPIN_DIR / TROYHIP_GTEST_SHIM_DIR   what cmake/reference_overlay uses to build the reference's own GPU tests against this package
get_filename_component(TROYHIP_ROOT "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(TROYHIP_INCLUDE_DIR "${TROYHIP_ROOT}/include")
set(TROYHIP_DROPIN_DIR "${TROYHIP_ROOT}/include/dropin")
set(TROYHIP_GTEST_SHIM_DIR "${TROYHIP_ROOT}/tests/cpp/gtest_shim")
if(NOT TROYHIP_LIBRARY)
  set(TROYHIP_LIBRARY "${TROYHIP_ROOT}/troy_amd/libtroyhip.so")
endif()
if(NOT EXISTS "${TROYHIP_LIBRARY}")
  set(troyhip_FOUND FALSE)
  set(troyhip_NOT_FOUND_MESSAGE "libtroyhip.so not found at ${TROYHIP_LIBRARY}: build it with `make -C ${TROYHIP_ROOT}/troy_amd/csrc` (hipcc --offload-arch=gfx950) or pass -DTROYHIP_LIBRARY=...")
  return()
endif()
if(NOT TARGET troyhip::troyhip)
  add_library(troyhip::troyhip SHARED IMPORTED)
  set_target_properties(troyhip::troyhip PROPERTIES
    IMPORTED_LOCATION "${TROYHIP_LIBRARY}"
    IMPORTED_NO_SONAME TRUE
    INTERFACE_INCLUDE_DIRECTORIES "${TROYHIP_INCLUDE_DIR}"
    INTERFACE_COMPILE_FEATURES cxx_std_17)
  find_package(Threads QUIET)  # troyn_devices.hpp: a host thread per device
  if(Threads_FOUND)
    set_property(TARGET troyhip::troyhip APPEND PROPERTY INTERFACE_LINK_LIBRARIES Threads::Threads)
  endif()
endif()
set(troyhip_FOUND TRUE)
