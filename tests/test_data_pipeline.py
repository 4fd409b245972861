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
