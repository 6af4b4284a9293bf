"""Regenerates include/troyn_hestd.inc (DATA: the security-standard bit bounds and SEAL's default coefficient moduli) from the
reference's tables src/utils/hestdparams.h and src/utils/globals.cpp.  Build container only (/root/reference)."""
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src = open(REF + "/src/utils/globals.cpp").read()
hs = open(REF + "/src/utils/hestdparams.h").read()
tab, bits = [], {}
for sec in (128, 192, 256):
    body = re.search(r"GetDefaultCoeffModulus%d\(\)(.*?)return default_coeff_modulus_%d;" % (sec, sec), src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    for n, v in re.findall(r"\{\s*(\d+)\s*,\s*\{([^}]*)\}\s*\}", body):
        tab.append((sec, int(n), [x.strip() for x in v.split(",") if x.strip()]))
    fn = re.search(r"seal_he_std_parms_%d_tc\(.*?\{(.*?)return 0;\s*\}" % sec, hs, re.S).group(1)
    bits[sec] = [int(b) for _, b in re.findall(r"size_t\((\d+)\):\s*return (\d+);", fn)]
out = []
for sec in (128, 192, 256):
    out.append("%d bits: %s" % (sec, bits[sec]))
    for s, n, vals in tab:
        if s == sec:
            out.append("  N=%d: %s" % (n, " ".join(vals)))
print("\n".join(out))
print("# compare with include/troyn_hestd.inc (tests/test_dropin.py does, when the reference is present)")
