"""The decoder oracle must reproduce the vectors captured from the REFERENCE's own
forward_mask / get_pos_encs (tests/golden/make_goldens.py) to <= 1e-5."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, arrays_to_clicks, golden_cases, load_case
from oracle import decoder as odec


@pytest.mark.parametrize("name", golden_cases())
def test_forward_mask_matches_reference(name, decoder_weights):
    c = load_case(name)
    ci, ct = arrays_to_clicks(c["click_rows"], c["click_objs"], c["click_times"], int(c["K"]))
    feats, xyz = torch.from_numpy(c["feats128"]), torch.from_numpy(c["xyz"])
    pos = odec.fourier_pos_enc(xyz, decoder_weights["pos_enc.gauss_B"], xyz.min(0)[0], xyz.max(0)[0])
    assert np.abs(pos.numpy() - c["pos_enc"]).max() <= 1e-5
    outs, masks = odec.forward_mask(decoder_weights, feats, xyz, pos, ci, ct, return_masks=True)
    for i in range(3):
        ref = c[f"logits{i}"]
        assert outs[i].shape == ref.shape
        err = np.abs(outs[i].numpy() - ref).max()
        assert err <= 1e-5 * max(1.0, np.abs(ref).max()), (name, i, err)
    for i in range(2):
        assert np.array_equal(masks[i].numpy(), c[f"attn_mask{i}"])


def test_goldens_cover_the_empty_label_rule():
    hit = False
    for name in golden_cases():
        h = load_case(name)["label_hist"]
        hit |= bool((h[:2] == 0).any())
    assert hit, "no golden case exercises agile3d.py:369,375 (all-True mask row reset)"


def test_time_table_shape():
    t = odec.time_table(128, 200)
    assert t.shape == (200, 128) and float(t[0, 1]) == 1.0 and float(t[0, 0]) == 0.0


def test_state_dict_key_layout(full_model_cpu):
    """Our parameter tree == the reference's (keys and shapes recorded from the reference model
    in the build container)."""
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    ours = {k: list(v.shape) for k, v in full_model_cpu.state_dict().items()}
    assert ours == ref
    assert sum(p.numel() for p in full_model_cpu.parameters()) == 39_289_760
    assert sum(p.numel() for p in full_model_cpu.backbone.parameters()) == 37_854_112
