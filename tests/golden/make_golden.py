#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

1. whisk_kat.json — the known-answer vectors the reference's own tests hold
   (/root/reference/src/whisk.rs:355-368, :381-402, :416-456): pure data (hex strings),
   extracted here so the tests can run where /root/reference does not exist.
2. oracle_vectors.json — vectors produced by the KAT-pinned CPU oracle (oracle/) for the
   BASELINE.json configs that the reference holds no vector for (ell = 28, 252, 1020):
   sha256 of the serialized proof + instance, so that GPU-side tests and later rounds can
   detect any drift of either implementation.
"""
import ctypes, hashlib, json, os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def extract_reference_kats():
    src = open("/root/reference/src/whisk.rs").read()
    out = {
        "source": "asn-d6/curdleproofs src/whisk.rs (tests serde_fr_rand, serde_g1_roundtrip, "
                  "whisk_tracker_proof, whisk_shuffle_proof)",
        "fr_le_bytes": re.search(r'hex::decode\("([0-9a-f]{64})"\)', src).group(1),
        "g1_generator_compressed": re.search(r'hex::decode\("([0-9a-f]{96})"\)', src).group(1),
        "tracker_proof": re.search(r'hex::encode\(tracker_proof\), "([0-9a-f]+)"', src).group(1),
        "whisk_shuffle_proof_ell124": re.search(
            r'hex::encode\(&whisk_shuffle_proof_bytes\), "([0-9a-f]+)"', src).group(1),
    }
    assert len(out["tracker_proof"]) == 2 * 128 and len(out["whisk_shuffle_proof_ell124"]) == 2 * 4496
    return out


def oracle_vectors():
    from tests.oracle_lib import Oracle
    orc = Oracle()
    vecs = {}
    for ell in (28, 60, 124, 252, 1020):
        crs = orc.generate_crs_points(ell)
        inst = orc.make_instance(ell, 0, crs)
        assert inst["verdict"] == 1
        vecs[str(ell)] = {
            "seed": 0,
            "proof_len": len(inst["proof"]),
            "proof_sha256": hashlib.sha256(inst["proof"]).hexdigest(),
            "crs_sha256": hashlib.sha256(crs).hexdigest(),
            "instance_sha256": hashlib.sha256(
                inst["vec_R"] + inst["vec_S"] + inst["vec_T"] + inst["vec_U"] + inst["M"]).hexdigest(),
            "prover_rand_sha256": hashlib.sha256(inst["prover_rand"]).hexdigest(),
            "proof_head_hex": inst["proof"][:96].hex(),
        }
    return vecs


if __name__ == "__main__":
    with open(os.path.join(HERE, "whisk_kat.json"), "w") as f:
        json.dump(extract_reference_kats(), f, indent=1)
    if "--kat-only" not in sys.argv:
        with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
            json.dump(oracle_vectors(), f, indent=1)
    print("ok")
