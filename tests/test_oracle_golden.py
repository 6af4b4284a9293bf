"""The CPU oracle against golden outputs of the REAL reference (tests/golden/*, produced by gen_golden.py from
oracle/_ref).  This is what pins the oracle; GPU parity is then measured against the oracle and the same files."""
import os

import numpy as np
import pytest

import cases
from conftest import GOLDEN


def _check_params(be, impl, gp):
    assert [str(p) for p in be.primes] == gp["primes"]
    assert str(impl.t) == gp["plain_modulus"]
    assert list(impl.chain()) == gp["chain"]
    for limbs, lv in gp["levels"].items():
        bsk, gamma = impl.behz_bases(int(limbs))
        assert [str(x) for x in bsk] == lv["bsk"] and str(gamma) == lv["gamma"]
    for i, p in enumerate(be.primes):
        t, g = impl.ntt_tables(i), gp["tables"][str(p)]
        assert str(t["root"]) == g["root"]
        assert [str(x) for x in t["inv_degree"]] == g["inv_degree"]
        for k in ("root_op", "root_quo", "inv_op", "inv_quo"):
            assert cases.sha(t[k]) == g[k], (p, k)


@pytest.mark.parametrize("name", cases.SMALL + cases.MEDIUM)
def test_oracle_scenario_matches_reference(name, golden_hashes, golden_params, oracle_lib):
    cfg = cases.CONFIGS[name]
    be = cases.oracle_backend(cfg)
    _check_params(be, be.impl, golden_params[name])
    out = cases.scenario(be, cfg)
    exp = golden_hashes[name]
    assert set(out) == set(exp)
    for k, m in out.items():
        e = exp[k]
        assert list(m.data.shape) == e["shape"], k
        assert cases.sha(m.data) == e["sha256"], k
        assert m.is_ntt == e["is_ntt"] and m.cf == e["cf"] and abs(m.scale - e["scale"]) <= 1e-12 * abs(e["scale"]), k


def test_oracle_full_limbs_n64(oracle_lib):
    name = "bfv_n64_k3"
    full = np.load(os.path.join(GOLDEN, f"golden_full_{name}.npz"))
    out = cases.scenario(cases.oracle_backend(cases.CONFIGS[name]), cases.CONFIGS[name])
    assert set(out) == set(full.files)
    for k in full.files:
        assert np.array_equal(out[k].data, full[k]), k


@pytest.mark.parametrize("name", cases.LARGE)
def test_oracle_headline_configs(name, golden_hashes, oracle_lib):
    """BFV N=2^15 L=14 multiply+relinearize (the BASELINE metric) and the CKKS N=2^15 chain, against the reference"""
    cfg = cases.CONFIGS[name]
    out = cases.scenario(cases.oracle_backend(cfg), cfg, light=True)
    for k, m in out.items():
        assert cases.sha(m.data) == golden_hashes[name][k]["sha256"], k


def test_cfgA_add_decrypt(oracle_lib):
    """BASELINE config A: BFV N=4096, 3 primes, encrypt -> add -> decrypt on the CPU path"""
    from oracle import oracle, ref
    f = np.load(os.path.join(GOLDEN, "cfgA_bfv_n4096_k3.npz"))
    primes, t = [int(x) for x in f["primes"]], int(f["t"])
    O = oracle.Oracle(oracle.BFV, 4096, primes, t)
    c1, c2 = ref.Ct(f["ct1"]), ref.Ct(f["ct2"])
    s = O.eval(ref.OP_ADD, c1, c2)
    assert cases.sha(s.data) == str(f["sum_sha256"])
    assert np.array_equal(O.decrypt(s, f["secret_key"]), f["decrypted"])
    assert np.array_equal(O.decrypt(c1, f["secret_key"]), f["plain1"])


@pytest.mark.parametrize("nm", ["bfv", "bgv", "ckks"])
def test_realkey_chain(nm, oracle_lib):
    """real keys from the reference's KeyGenerator: multiply -> relinearize (-> rescale) -> rotate, bit-exact + decrypts"""
    from oracle import oracle, ref
    f = np.load(os.path.join(GOLDEN, f"realkey_{nm}.npz"))
    scheme = dict(bfv=oracle.BFV, bgv=oracle.BGV, ckks=oracle.CKKS)[nm]
    primes, t = [int(x) for x in f["primes"]], int(f["t"])
    O = oracle.Oracle(scheme, 128, primes, t)
    O.set_kswitch_key(0, f["relin_key"])
    O.set_kswitch_key(int(f["galois_elt"]), f["galois_key"])
    if nm == "ckks":
        c1, c2 = ref.Ct(f["ct1"], True, float(f["in_scale"])), ref.Ct(f["ct2"], True, float(f["in_scale"]))
        r = O.eval(ref.OP_RELIN, O.eval(ref.OP_MULTIPLY, c1, c2))
        r = O.eval(ref.OP_ROTATE_VECTOR, O.eval(ref.OP_RESCALE_NEXT, r), iarg=1)
        assert np.array_equal(r.data, f["result"]) and r.scale == float(f["result_scale"])
    else:
        c1, c2 = ref.Ct(f["ct1"], False, 1.0, int(f["ct1_cf"])), ref.Ct(f["ct2"], False, 1.0, int(f["ct2_cf"]))
        r = O.eval(ref.OP_ROTATE_ROWS, O.eval(ref.OP_RELIN, O.eval(ref.OP_MULTIPLY, c1, c2)), iarg=1)
        assert np.array_equal(r.data, f["result"]) and r.correction_factor == int(f["result_cf"])
        assert np.array_equal(O.decrypt(r, f["secret_key"]), f["decrypted"])


@pytest.mark.parametrize("name", cases.SIZES)
def test_oracle_general_sizes(name, oracle_lib):
    """3x2 / 3x3 / 2x3 multiply, size-3 square, relinearize 4 -> 2 and 5 -> 2 (reference: src/evaluator.cpp:385-794, 1113-1163;
    note the reference never advances its `encrypted_iter`): the oracle against the reference's outputs (golden_sizes.json)"""
    import json
    exp = json.load(open(os.path.join(GOLDEN, "golden_sizes.json")))[name]
    cfg = cases.CONFIGS[name]
    out = cases.scenario_sizes(cases.oracle_backend(cfg), cfg)
    assert set(out) == set(exp)
    for k, m in out.items():
        assert cases.sha(m.data) == exp[k]["sha256"] and list(m.data.shape) == exp[k]["shape"], k
        assert m.is_ntt == exp[k]["is_ntt"] and m.cf == exp[k]["cf"] and abs(m.scale - exp[k]["scale"]) <= 1e-12 * abs(exp[k]["scale"])


@pytest.mark.parametrize("name", cases.CHAIN)
def test_oracle_ckks_chain_depth3(name, oracle_lib):
    """multiply -> relinearize -> rescale -> rotate(1) chained to depth 3 (BASELINE configs[2] as SURVEY.md 8d states it), limbs after
    EVERY op against the reference's own outputs (golden_chain.json), incl. N = 2^15 K = 15"""
    import json
    exp = json.load(open(os.path.join(GOLDEN, "golden_chain.json")))[name]
    cfg = cases.CONFIGS[name]
    out = cases.scenario_chain(cases.oracle_backend(cfg), cfg)
    assert set(out) == set(exp)
    for k, m in out.items():
        assert cases.sha(m.data) == exp[k]["sha256"], k
