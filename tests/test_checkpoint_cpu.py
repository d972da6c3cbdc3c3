"""Checkpoint save / load of the model facade (reference: models/base_model.py:33-112) -- host logic only, runs without a GPU."""
import os

import pytest
import torch

from ppst_amd import weights as W
from ppst_amd.ppst_model import Options, PPSTModel


def _model(tmp, **kw):
    opt = Options(checkpoints_dir=str(tmp), name="exp", **kw)
    return PPSTModel(opt, with_D=True)


def test_save_then_load_round_trip(tmp_path):
    sd = W.make_state_dict(7, with_D=True, with_nce=True, bias_std=0.1, noise_weight=0.1)
    m = _model(tmp_path, isTrain=True)
    m.load_weights(sd)
    path = m.save(12000)
    assert os.path.basename(path) == "12k_checkpoint.pth"
    link = os.path.join(str(tmp_path), "exp", "latest_checkpoint.pth")
    assert os.path.islink(link) and os.readlink(link) == "12k_checkpoint.pth"
    m2 = _model(tmp_path, isTrain=True)
    assert m2.load(verbose=False)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    # a second save replaces the symlink
    m.save(13000)
    assert os.readlink(link) == "13k_checkpoint.pth"


def test_test_mode_skips_discriminator_and_requires_file(tmp_path):
    sd = W.make_state_dict(8, with_D=True, with_nce=True)   # the reference checkpoint also holds criterionNCE.* buffers
    os.makedirs(os.path.join(str(tmp_path), "exp"))
    torch.save(sd, os.path.join(str(tmp_path), "exp", "latest_checkpoint.pth"))
    m = _model(tmp_path, isTrain=False)
    d_before = {k: v.clone() for k, v in m.state_dict().items() if k.startswith("D.")}
    assert m.load(verbose=False)
    own = m.state_dict()
    for k in own:
        if k.startswith("D."):
            assert torch.equal(own[k], d_before[k]), "D.* must be left alone at test time: %s" % k
        else:
            assert torch.equal(own[k], sd[k]), k
    missing = _model(tmp_path, isTrain=False, resume_iter="50k")
    with pytest.raises(AssertionError):
        missing.load(verbose=False)
    assert _model(tmp_path, isTrain=True, resume_iter="50k").load(verbose=False) is False


def test_missing_keys_are_skipped_and_shape_mismatch_is_loud(tmp_path, capsys):
    sd = W.make_state_dict(9, with_D=True, with_nce=False)
    drop = "G.ToRGB.bias" if "G.ToRGB.bias" in sd else [k for k in sd if k.startswith("G.")][0]
    bad = [k for k in sd if k.startswith("E1.") and sd[k].dim() == 4][0]
    part = {k: v for k, v in sd.items() if k != drop}
    part[bad] = torch.ones(tuple(d + 1 for d in sd[bad].shape))
    os.makedirs(os.path.join(str(tmp_path), "exp"))
    torch.save(part, os.path.join(str(tmp_path), "exp", "latest_checkpoint.pth"))
    m = _model(tmp_path, isTrain=True)
    with pytest.raises(ValueError, match="Shape does not match"):
        m.load(verbose=False)
    m = _model(tmp_path, isTrain=True)
    before = m.state_dict()[drop].clone()
    assert m.load(force="all")
    assert "Key %s does not exist in checkpoint" % drop in capsys.readouterr().out
    assert torch.equal(m.state_dict()[drop], before)
    assert torch.equal(m.state_dict()[bad], torch.ones_like(sd[bad]))  # overlapping corner copied (reference "all")
