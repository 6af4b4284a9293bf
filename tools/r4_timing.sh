#!/bin/bash
# phase stamps of one workgroup of the single-pass forward kernel (-DN1_TIMING build, tools/probe_libs/libtroyhip_timing.so): shader-clock deltas per row
# between: row start | round A | barrier | staging + barrier | round B | round C1 | (C2 + stores) | barrier | staging | B | C1 | C2 + stores | next row
export TROYHIP_LIB=$PWD/tools/probe_libs/libtroyhip_timing.so
for bits in "[49] * 15" "[60] + [58] * 13 + [60]"; do echo "== $bits"; PROBE_BITS="$bits" python tools/ntt1_probe.py 128 2 2>&1 | tail -9; done
