#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the REAL reference CPU path (oracle/_ref, built from
/root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/gen_golden.py

Outputs (data only -- inputs are regenerated from seeds by troy_amd.synth, see tests/cases.py):
  golden_full_<cfg>.npz   every scenario output limb-for-limb (small N)
  golden_hashes.json      SHA-256 of every scenario output + metadata, for all configs incl. N=32768
  golden_params.json      primes, plain modulus, BEHZ bases, psi, table hashes per config
  golden_chain.json       SHA-256 after every op of the CKKS multiply -> relinearize -> rescale -> rotate chain at depth 3 (cases.scenario_chain)
  golden_sizes.json       SHA-256 of the general-size scenario (3x2 / 3x3 multiply, relinearize 4->2 / 5->2), cases.scenario_sizes
  cfgA_bfv_n4096_k3.npz   BASELINE config A: seeded keys, two encryptions, their sum, the decryption
  realkey_<scheme>.npz    one decrypt-verified real-key case per scheme (N=128)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..")))

import cases  # noqa: E402
from oracle import ref  # noqa: E402


def meta_dict(m):
    return dict(sha256=cases.sha(m.data), shape=list(m.data.shape), is_ntt=m.is_ntt, scale=m.scale, cf=m.cf)


def sizes_only():
    sizes = {}
    for name in cases.SIZES:
        cfg = cases.CONFIGS[name]
        sizes[name] = {k: meta_dict(v) for k, v in cases.scenario_sizes(cases.ref_backend(cfg), cfg).items()}
        print("sizes", name)
    json.dump(sizes, open(os.path.join(HERE, "golden_sizes.json"), "w"), indent=1, sort_keys=True)
    chain_only()


def chain_only():
    out = {}
    for name in cases.CHAIN:
        cfg = cases.CONFIGS[name]
        out[name] = {k: meta_dict(v) for k, v in cases.scenario_chain(cases.ref_backend(cfg), cfg).items()}
        print("chain", name)
    json.dump(out, open(os.path.join(HERE, "golden_chain.json"), "w"), indent=1, sort_keys=True)


def main():
    if "--sizes-only" in sys.argv:  # same content as the full run writes into golden_sizes.json
        return sizes_only()
    if "--chain-only" in sys.argv:  # ... golden_chain.json
        return chain_only()
    hashes, params, sizes = {}, {}, {}
    for name in cases.SMALL + cases.MEDIUM + cases.LARGE:
        cfg = cases.CONFIGS[name]
        be = cases.ref_backend(cfg)
        R = be.impl
        light = name in cases.LARGE
        out = cases.scenario(be, cfg, light=light)
        hashes[name] = {k: meta_dict(v) for k, v in out.items()}
        if name in cases.SIZES:
            sizes[name] = {k: meta_dict(v) for k, v in cases.scenario_sizes(cases.ref_backend(cfg), cfg).items()}
        K = len(be.primes)
        p = dict(primes=[str(x) for x in be.primes], plain_modulus=str(R.t), chain=list(R.chain()), levels={}, tables={})
        for limbs in range(R.chain()[2], K + 1):
            bsk, gamma = R.behz_bases(limbs)
            p["levels"][str(limbs)] = dict(bsk=[str(x) for x in bsk], gamma=str(gamma))
        for i in range(K):
            t = R.ntt_tables(i)
            p["tables"][str(be.primes[i])] = dict(root=str(t["root"]), inv_degree=[str(x) for x in t["inv_degree"]],
                                                  root_op=cases.sha(t["root_op"]), root_quo=cases.sha(t["root_quo"]),
                                                  inv_op=cases.sha(t["inv_op"]), inv_quo=cases.sha(t["inv_quo"]),
                                                  head=[str(x) for x in t["root_op"][:4]])
        params[name] = p
        if name == "bfv_n64_k3":
            np.savez_compressed(os.path.join(HERE, f"golden_full_{name}.npz"), **{k: v.data for k, v in out.items()})
        print("done", name, len(out), "outputs")
    json.dump(hashes, open(os.path.join(HERE, "golden_hashes.json"), "w"), indent=1, sort_keys=True)
    json.dump(params, open(os.path.join(HERE, "golden_params.json"), "w"), indent=1, sort_keys=True)
    json.dump(sizes, open(os.path.join(HERE, "golden_sizes.json"), "w"), indent=1, sort_keys=True)
    chain_only()

    # ---- cfgA: BFV N=4096 K=3 encrypt -> add -> decrypt on the reference CPU path
    cfg = cases.CONFIGS["cfgA_bfv_n4096_k3"]
    primes = ref.coeff_modulus_create(cfg["N"], cfg["bits"])
    t = ref.plain_batching(cfg["N"], cfg["tbits"])
    R = ref.Ref(ref.BFV, cfg["N"], primes, t, seed=20261001)
    R.keygen([])
    rng = np.random.default_rng(7)
    v1 = rng.integers(0, t, cfg["N"], dtype=np.uint64)
    v2 = rng.integers(0, t, cfg["N"], dtype=np.uint64)
    p1, p2 = R.batch_encode(v1), R.batch_encode(v2)
    c1, c2 = R.encrypt(p1), R.encrypt(p2)
    s = R.eval(ref.OP_ADD, c1, c2)
    d, budget = R.decrypt(s)
    assert np.array_equal(R.batch_decode(d), (v1 + v2) % np.uint64(t))
    np.savez_compressed(os.path.join(HERE, "cfgA_bfv_n4096_k3.npz"), primes=np.array(primes, dtype=np.uint64), t=np.uint64(t),
                        secret_key=R.secret_key(), public_key=R.public_key(), values1=v1, values2=v2, plain1=p1, plain2=p2,
                        ct1=c1.data, ct2=c2.data, sum_sha256=np.array(cases.sha(s.data)), decrypted=d)
    print("cfgA budget", budget)

    # ---- one real-key, decrypt-verified case per scheme at N=128
    for scheme, bits, tbits, nm in ((ref.BFV, [40, 40, 40, 40], 10, "bfv"), (ref.BGV, [40, 36, 36, 40], 10, "bgv"), (ref.CKKS, [40, 30, 30, 40], 0, "ckks")):
        N = 128
        primes = ref.coeff_modulus_create(N, bits)
        t = ref.plain_batching(N, tbits) if scheme != ref.CKKS else 0
        R = ref.Ref(scheme, N, primes, t, seed=99)
        e1 = R.elt_from_step(1)
        R.keygen([e1])
        store = dict(primes=np.array(primes, dtype=np.uint64), t=np.uint64(t), secret_key=R.secret_key(), relin_key=R.relin_key(),
                     galois_key=R.galois_key(e1), galois_elt=np.uint32(e1))
        if scheme != ref.CKKS:
            v1 = rng.integers(0, t, N, dtype=np.uint64)
            v2 = rng.integers(0, t, N, dtype=np.uint64)
            c1, c2 = R.encrypt(R.batch_encode(v1)), R.encrypt(R.batch_encode(v2))
            r = R.eval(ref.OP_RELIN, R.eval(ref.OP_MULTIPLY, c1, c2))
            r = R.eval(ref.OP_ROTATE_ROWS, r, iarg=1)
            d, budget = R.decrypt(r)
            half = N // 2
            prod = (v1 * v2) % np.uint64(t)
            expect = np.concatenate([np.roll(prod[:half], -1), np.roll(prod[half:], -1)])
            assert np.array_equal(R.batch_decode(d), expect), nm
            store.update(ct1=c1.data, ct2=c2.data, ct1_cf=np.uint64(c1.correction_factor), ct2_cf=np.uint64(c2.correction_factor),
                         result=r.data, result_cf=np.uint64(r.correction_factor), decrypted=d, values1=v1, values2=v2)
        else:
            scale = 2.0 ** 30
            z1 = rng.normal(size=N // 2) + 1j * rng.normal(size=N // 2)
            z2 = rng.normal(size=N // 2) + 1j * rng.normal(size=N // 2)
            c1 = R.ckks_encrypt(R.ckks_encode(z1, scale, 3), scale)
            c2 = R.ckks_encrypt(R.ckks_encode(z2, scale, 3), scale)
            r = R.eval(ref.OP_RELIN, R.eval(ref.OP_MULTIPLY, c1, c2))
            r = R.eval(ref.OP_RESCALE_NEXT, r)
            r = R.eval(ref.OP_ROTATE_VECTOR, r, iarg=1)
            dec = R.ckks_decrypt_decode(r)
            assert np.max(np.abs(dec - np.roll(z1 * z2, -1))) < 1e-2, np.max(np.abs(dec - np.roll(z1 * z2, -1)))
            store.update(ct1=c1.data, ct2=c2.data, in_scale=np.float64(scale), result=r.data, result_scale=np.float64(r.scale),
                         slots1=z1, slots2=z2, decoded=dec)
        np.savez_compressed(os.path.join(HERE, f"realkey_{nm}.npz"), **store)
        print("realkey", nm, "ok")


if __name__ == "__main__":
    main()
