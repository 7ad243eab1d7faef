#!/usr/bin/env python3
"""Lane-length model of the bucket-list MSM waves (msm_body.hpp `msm_tblw_body`): how long the slowest lane of a wave walks, against the mean.

A wave sorts its (point, window) entries into 128 buckets by |digit|, ranks the buckets by size and gives lane l the buckets of rank l and
127 - l; the wave's loop runs until its slowest lane is through, so  mean / max  of the lanes' entry counts bounds what fraction of the
addition rate the loop can reach (everything else — digit loads, histogram, scatter, ranking, parking — comes on top).  Digits are the
signed radix-256 digits of uniformly random scalars: |d| = 1..127 with probability 2/256 each, 128 and 0 with 1/256.

    python scripts/bucket_lane_histogram.py [waves per shape]

Shapes: the prover's table MSMs (k_msm_tblw<32, false>: 128 points x 32 windows, and 256 x 32), the verifier's per-proof points
(k_msm_tblw<2, true>: 1112 points x 2 windows; config 5 groups: rounds of 4096 points), R and S (k_msm_tblw_pair: 252 points x 2 windows,
two tasks on one digit sort: four lists per lane, the second task's pair taken from the partner lane of rank 63 - r)."""
import sys
import numpy as np


def bucket_counts(rng, waves, entries):
    # magnitudes 1..128 (index 0..127) and the zero digit (index 128, dropped)
    p = np.array([2.0] * 127 + [1.0] + [1.0]) / 256.0
    return rng.multinomial(entries, p, size=waves)[:, :128]


def single(rng, waves, points, windows):
    c = np.sort(bucket_counts(rng, waves, points * windows), axis=1)
    lanes = c[:, :64] + c[:, ::-1][:, :64]          # rank l with rank 127 - l
    return lanes


def pair(rng, waves, points, windows):
    lanes = single(rng, waves, points, windows)      # the same lists serve both tasks
    r = np.sort(lanes, axis=1)
    return r + r[:, ::-1]                            # own pair + the pair of the lane of rank 63 - r


def report(name, lanes, fixed_adds=0.0):
    mean, mx = lanes.mean(axis=1), lanes.max(axis=1)
    eff = mean / mx
    h = np.bincount((lanes - lanes.mean(axis=1, keepdims=True).round().astype(int)).clip(-20, 20).ravel() + 20, minlength=41)
    print("%-58s mean %7.2f  longest lane %7.2f (p50 %5.0f, p99 %5.0f)  mean/longest %.3f" % (name, mean.mean(), mx.mean(), np.percentile(mx, 50), np.percentile(mx, 99), eff.mean()))
    tot = h.sum()
    print("    lane length - wave mean, share of lanes: " + " ".join("%+d:%.1f%%" % (k - 20, 100.0 * v / tot) for k, v in enumerate(h) if v / tot >= 0.005))


def main():
    waves = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(6)
    report("k_msm_tblw<32, false>  128 points x 32 windows", single(rng, waves, 128, 32))
    report("k_msm_tblw<32, false>  256 points x 32 windows (B_t, B_u)", single(rng, waves, 256, 32))
    report("k_msm_tblw<2, true>    1112 points x 2 windows (verifier)", single(rng, waves, 1112, 2))
    report("k_msm_tblw<2, true>    4096 points x 2 windows (config 5 round)", single(rng, waves, 4096, 2))
    report("k_msm_tblw_pair        252 points x 2 windows, two tasks", pair(rng, waves, 252, 2))
    report("  (the same without the partner lane: own pair twice)", 2 * single(rng, waves, 252, 2))


if __name__ == "__main__":
    main()
