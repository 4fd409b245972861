"""The input step in front of the path, host side: LFR stacking and AudioDataset's batching against the
reference's outputs (tests/golden/g10_input.npz: utils/data.py:191-218 and :28-110, the latter also on the reference's own
test/data/data.json)."""
import json
import os

import numpy as np

from asr_amd import data


def test_lfr_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "g10_input.npz"))
    n = 0
    for k in z.files:
        if k.startswith("lfr_T"):
            T, m, nn = (int(p[1:]) for p in k.split("_")[1:])
            got = data.build_LFR_features(z["lfr_x_T%d" % T], m, nn)
            np.testing.assert_array_equal(got, z[k])
            n += 1
    assert n == 25


def test_minibatch_plans_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "g10_input.npz"))
    plans = json.loads(str(z["batch_plans"]))
    cfgs = {"count": dict(batch_size=8, max_length_in=800, max_length_out=30),
            "frames": dict(batch_size=8, max_length_in=800, max_length_out=30, batch_frames=2000),
            "first2": dict(batch_size=5, max_length_in=400, max_length_out=150, num_batches=2)}
    for name in ("syn", "ref"):
        utts = json.loads(str(z["batch_utts_" + name]))
        for tag, kw in cfgs.items():
            got = [[k for k, _ in mb] for mb in data.make_minibatches(utts, **kw)]
            assert got == plans["%s_%s" % (name, tag)], (name, tag)
    # the T / U < 5 utterance never shows up; the equal-length pair keeps its file order
    flat = [k for mb in data.make_minibatches(json.loads(str(z["batch_utts_syn"])), 8, 800, 30) for k, _ in mb]
    assert "utt900" not in flat
    if "utt003" in flat:
        assert flat.index("utt003") + 1 == flat.index("utt901")


def test_shard_by_length_balances_frames():
    rng = np.random.default_rng(0)
    lens = rng.integers(100, 1600, 256).tolist()
    shards = data.shard_by_length(lens, 8)
    assert sorted(i for s in shards for i in s) == list(range(256)) and all(len(s) == 32 for s in shards)
    totals = [sum(lens[i] for i in s) for s in shards]
    assert (max(totals) - min(totals)) / np.mean(totals) < 0.02          # contiguous slices of the sorted list would differ by > 2x
    naive = [sum(sorted(lens, reverse=True)[r * 32:(r + 1) * 32]) for r in range(8)]
    assert max(naive) / min(naive) > 2
