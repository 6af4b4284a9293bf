#!/usr/bin/env python3
"""isa_census.py -- instruction census of the gfx950 code objects, per kernel and per loop, from `hipcc -S` output.

    tools/isa_census.py troy_amd/csrc/ntt1.hip [--kernel SUBSTR] [--loops]        (compiles with the Makefile's flags)
    tools/isa_census.py build/asm/ntt1.s

Classes: VALU (all v_* except the memory classes), of which MUL = the multiply class (v_mad_u64_u32, v_mul_lo/hi_u32, v_mul_*_u24,
v_*_f64 multiplies / fmas), CARRY (v_add_co/addc/sub_co/subb and their rev forms), CND (v_cndmask), MOV (v_mov / v_accvgpr / readlane);
LDS (ds_*), VMEM (global_/buffer_/flat_/scratch_), SALU (s_* without waitcnt / barrier / branch), SMEM (s_load*, s_buffer_load*),
WAIT (s_waitcnt), BAR (s_barrier), BR (branches).  A loop = [label, backward branch to that label]; nested loops are reported with
their own bodies, and a loop's "own" count excludes the loops nested in it.  The per-row dynamic count of the row loops is the number
the DESIGN.md tables quote.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

HIPFLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -S".split()

MULS = ("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u24",
        "v_mul_f64", "v_fma_f64", "v_mfma")
CARRY = ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32")


def classify(op):
    if op.startswith("v_"):
        if op.startswith(MULS):
            return "MUL"
        if op.startswith(CARRY):
            return "CARRY"
        if op.startswith("v_cndmask"):
            return "CND"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane")):
            return "MOV"
        return "VOTHER"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith("s_barrier"):
        return "BAR"
    if op.startswith(("s_cbranch", "s_branch")):
        return "BR"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


VALU_CLASSES = ("MUL", "CARRY", "CND", "MOV", "VOTHER")


def parse(path):
    kernels = collections.OrderedDict()
    cur, name = None, None
    meta = {}
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith("\t"):
            name = m.group(1)
            cur = kernels.setdefault(name, [])
            continue
        if cur is None:
            continue
        s = ln.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") or s.startswith(".rodata"):
            pass
        m = re.match(r"^\.(?:L|LBB)(\w+):", s)
        if m:
            cur.append(("label", s.split(":")[0]))
            continue
        if s.startswith("s_endpgm"):
            cur.append(("inst", "s_endpgm", s))
            cur = None
            continue
        if not s or s.startswith((";", ".", "//")):
            mm = re.match(r"^;\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize|codeLenInByte):\s*(\d+)", s)
            if mm and name:
                meta.setdefault(name, {})[mm.group(1)] = int(mm.group(2))
            continue
        op = s.split()[0]
        cur.append(("inst", op, s))
    # trailing metadata comments come after s_endpgm: second sweep
    name = None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name = m.group(1)
        mm = re.match(r"^;\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize|codeLenInByte):\s*(\d+)", ln.strip())
        if mm and name:
            meta.setdefault(name, {})[mm.group(1)] = int(mm.group(2))
    return kernels, meta


def census(items):
    c = collections.Counter()
    for it in items:
        if it[0] == "inst":
            c[classify(it[1])] += 1
    return c


def loops_of(items):
    """[(start_idx, end_idx, label)] for every backward branch"""
    pos = {}
    out = []
    for i, it in enumerate(items):
        if it[0] == "label":
            pos[it[1]] = i
        elif it[0] == "inst" and it[1].startswith(("s_cbranch", "s_branch")):
            tgt = it[2].split()[-1]
            if tgt in pos:
                out.append((pos[tgt], i, tgt))
    # merge loops sharing a header: keep the widest
    best = {}
    for s, e, l in out:
        if l not in best or e > best[l][1]:
            best[l] = (s, e, l)
    return sorted(best.values())


def fmt(c):
    valu = sum(c[k] for k in VALU_CLASSES)
    return (f"VALU {valu:5d} (mul {c['MUL']:4d} carry {c['CARRY']:4d} cnd {c['CND']:4d} mov {c['MOV']:4d} other {c['VOTHER']:4d})  "
            f"LDS {c['LDS']:4d} VMEM {c['VMEM']:4d} SALU {c['SALU']:4d} SMEM {c['SMEM']:3d} WAIT {c['WAIT']:3d} BAR {c['BAR']:2d} BR {c['BR']:3d}")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--kernel", default="", help="only kernels whose demangled name contains this")
    ap.add_argument("--loops", action="store_true", help="per-loop breakdown")
    ap.add_argument("--flags", default="", help="extra hipcc flags, e.g. -DN2_MAC_WAVES=3")
    a = ap.parse_args()
    path = a.src
    if not path.endswith(".s"):
        tmp = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
        cmd = ["/opt/rocm/bin/hipcc"] + HIPFLAGS + a.flags.split() + [path, "-o", tmp]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        path = tmp
    kernels, meta = parse(path)
    dm = demangle(list(kernels))
    for k, items in kernels.items():
        nm = dm.get(k, k)
        nm = re.sub(r"^void troyhip::", "", nm)
        nm = re.sub(r"\(troyhip::\w+\)$", "", nm)
        if a.kernel and a.kernel not in nm:
            continue
        if not any(it[0] == "inst" for it in items):
            continue
        md = meta.get(k, {})
        print(f"== {nm}   vgprs {md.get('NumVgprs', '?')} agprs {md.get('NumAgprs', '?')} scratch {md.get('ScratchSize', '?')} B occupancy {md.get('Occupancy', '?')} lds {md.get('LDSByteSize', '?')}")
        print("   static  " + fmt(census(items)))
        if a.loops:
            lp = loops_of(items)
            for (s, e, l) in lp:
                inner = [(s2, e2) for (s2, e2, _) in lp if s2 > s and e2 < e or (s2 >= s and e2 < e and (s2, e2) != (s, e))]
                own = [it for i, it in enumerate(items[s:e + 1], s) if not any(s2 <= i <= e2 for s2, e2 in inner)]
                depth = sum(1 for (s2, e2, _) in lp if s2 <= s and e2 >= e and (s2, e2) != (s, e))
                print(f"   {'  ' * depth}loop {l:12s} [{e - s + 1:5d} lines] own " + fmt(census(own)))


if __name__ == "__main__":
    main()
