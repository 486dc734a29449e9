"""Drop-in boundary B1: the product's ``TrainingModule`` exposes exactly the parameter / buffer
names and shapes of the reference's (``stp3/trainer.py:14-97`` + every submodule), so reference checkpoints load.

tests/golden/state_dict_keys.json was written by oracle/make_golden_train.py from the reference's own
``TrainingModule(cfg).state_dict()`` (BASELINE configs[2] overrides) in the build container."""
import json
import os

from stp3_amd.config import perception_cfg
from stp3_amd.trainer import TrainingModule
from tests import helpers as H

C3 = {'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True}


def test_training_module_state_dict_matches_reference():
    want = json.load(open(os.path.join(H.GOLDEN, 'state_dict_keys.json')))['TrainingModule_c3']
    cfg = perception_cfg(**{'LIFT.GT_DEPTH': True, 'INSTANCE_SEG.ENABLED': True, 'INSTANCE_FLOW.ENABLED': True})
    got = {k: list(v.shape) for k, v in TrainingModule(cfg.convert_to_dict()).state_dict().items()}
    missing = sorted(set(want) - set(got))
    extra = sorted(set(got) - set(want))
    assert not missing and not extra, (missing[:5], extra[:5])
    wrong = [k for k in want if want[k] != got[k]]
    assert not wrong, wrong[:5]
    assert len(want) > 850          # the whole model: trunk, heads, temporal model, decoder, loss weights


def test_load_from_checkpoint_and_pretrained_filter(tmp_path):
    """evaluate.py:31 ``TrainingModule.load_from_checkpoint(path, strict=True)`` and the ``'decoder' not in k`` filter
    of train.py:21-29, on a checkpoint in Lightning's layout written by the product itself."""
    import torch
    from stp3_amd.config import perception_cfg
    from stp3_amd.trainer import TrainingModule
    from tests import helpers as H
    src = TrainingModule(perception_cfg(**C3).convert_to_dict())
    H.fill_deterministic(src.model, seed=3)
    path = str(tmp_path / 'ckpt.pt')
    torch.save(src.checkpoint(), path)
    dst = TrainingModule.load_from_checkpoint(path, strict=True)
    assert dst.cfg.LIFT.GT_DEPTH and dst.cfg.INSTANCE_FLOW.ENABLED               # hyper-parameters restored
    a, b = src.state_dict(), dst.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    # pretrained initialisation: everything but the decoders
    fresh = TrainingModule(perception_cfg(**C3).convert_to_dict())
    H.fill_deterministic(fresh.model, seed=4)
    before = {k: v.clone() for k, v in fresh.state_dict().items()}
    loaded = fresh.load_pretrained_weights(path)
    after = fresh.state_dict()
    assert loaded and not any('decoder' in k for k in loaded)
    for k in after:
        want = a[k] if ('decoder' not in k) else before[k]
        assert torch.equal(after[k], want), k
