"""Generates tests/golden/golden_wire.json from the REAL reference (oracle/_ref, build container only): the 256-bit parms_id of every level
(EncryptionParameters::computeParmsID, src/encryptionparams.cpp:118-146) of the contexts tests/cpp/dump_wire.cpp serializes in.  The wire-format
test puts these ids -- not the product's own -- into the blobs it expects."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import ref as R  # noqa: E402

assert R.available(), "build oracle/_ref first: make -C oracle ref"
out = {}
for tag, scheme in (("bfv", 1), ("ckks", 2), ("bgv", 3)):
    N, bits = 64, [40, 40, 40]
    primes = R.coeff_modulus_create(N, bits)
    t = R.plain_batching(N, 17) if scheme != 2 else 0
    r = R.Ref(scheme, N, primes, t)
    key, first, last = r.chain()
    ids = {str(len(primes)): r.parms_id(len(primes))}
    for limbs in range(first, last - 1, -1):
        ids[str(limbs)] = r.parms_id(limbs)
    out[tag] = {"scheme": scheme, "N": N, "bits": bits, "primes": [int(p) for p in primes], "plain_modulus": int(t), "chain": [key, first, last], "parms_id": ids}
json.dump(out, open(os.path.join(HERE, "golden_wire.json"), "w"), indent=1)
print("wrote golden_wire.json:", {k: v["chain"] for k, v in out.items()})
