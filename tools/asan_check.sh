#!/bin/bash
# AddressSanitizer over the kernel sources, on the CPU: the emulator build of libtroyhip (tests/emul/hip_emul.h) compiled with
# -fsanitize=address, driven through the BEHZ kernels at every limb count, random parameter sets, the single-pass NTT, the two C++
# shim tests and the wire-format dumper.  LDS arrays are globals and device allocations are heap blocks there, so out-of-range indexing shows up.  (GPU sanitizers are
# not available on the pool.)  usage: tools/asan_check.sh
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C troy_amd/csrc emul OBJDIR=/tmp/troy_build_asan EMUL_OUT=/tmp/libtroyhip_emul_asan.so \
     EMUL_FLAGS="-O1 -g -fsanitize=address -fno-omit-frame-pointer" EMUL_LDFLAGS="-fsanitize=address"
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0   # ucontext fibers: no fake stacks
LD_PRELOAD=$(gcc -print-file-name=libasan.so) TROYHIP_NTT=single python tools/asan_run.py
# the launch shapes the library only takes for large grids, forced: the wide strided pass, the XCD-aware (grouped) workgroup order
LD_PRELOAD=$(gcc -print-file-name=libasan.so) TROYHIP_NTT=twopass TROYHIP_NTT2_WIDE=1 python tools/asan_run.py
LD_PRELOAD=$(gcc -print-file-name=libasan.so) TROYHIP_NTT=single TROYHIP_NTT1_XCD=1 TROYHIP_NTT1_XCD_GROUP=3 python tools/asan_run.py
# UndefinedBehaviorSanitizer over the same drive (default kernels and the alternative forms)
make -s -j8 -C troy_amd/csrc emul OBJDIR=/tmp/troy_build_ubsan EMUL_OUT=/tmp/libtroyhip_emul_ubsan.so \
     EMUL_FLAGS="-O1 -g -fsanitize=undefined -fno-sanitize-recover=undefined" EMUL_LDFLAGS="-fsanitize=undefined"
for e in TROYHIP_NTT=single TROYHIP_BEHZ=valu TROYHIP_NTT=twopass; do
    env $e TROY_EMUL_LIB=/tmp/libtroyhip_emul_ubsan.so LD_PRELOAD=$(gcc -print-file-name=libubsan.so) python tools/asan_run.py
done
for t in test_troyn test_troyn_app; do
    g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -Iinclude tests/cpp/$t.cpp -o /tmp/${t}_asan /tmp/libtroyhip_emul_asan.so -Wl,-rpath,/tmp
    /tmp/${t}_asan | grep -E "FAIL|ALL OK"
done
# the serializers and their loader fuzz (truncated streams, oversized size fields)
g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -Iinclude tests/cpp/dump_wire.cpp -o /tmp/dump_wire_asan /tmp/libtroyhip_emul_asan.so -Wl,-rpath,/tmp
mkdir -p /tmp/wire_asan && /tmp/dump_wire_asan /tmp/wire_asan | grep -E "FAIL|ALL OK"
