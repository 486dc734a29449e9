"""GPU: the planner on the MI355X -- csrc/stp3_plan.hip behind Cost_Function, against the
fixtures of the reference's own classes (tests/golden/planning.npz); Planning in float32 and under bf16 autocast; the
cost kernel on a full-size sample set (B=4, 1 800 trajectories, 6 steps) against the torch statements evaluated on the
same device; and one training step of a planning configuration end to end."""
import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_planning_cpu import PLANNING, cfg, check_costs, check_planner, planner_case, split_hd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('form', ['train', 'eval'])
def test_cost_kernel_matches_the_reference(form):
    check_costs(form, 'cuda')


def test_planner_float32():
    check_planner(planner_case('cuda'), tol=1e-4, gtol=5e-3)


def test_planner_bf16_autocast():
    """bf16 convolutions in the camera-feature reduction; costs, selection and the GRU stay float32."""
    out = planner_case('cuda', autocast=True)
    g = H.load('planning.npz')
    assert abs(out['loss'] - g['planner/train/loss'][0]) <= 2e-2 * abs(g['planner/train/loss'][0])
    assert np.abs(out['traj'].numpy() - g['planner/train/traj']).max() <= 5e-2
    np.testing.assert_allclose(out['d_cost_volume'].numpy(), g['planner/train/d_cost_volume'], rtol=1e-5, atol=1e-5)


def test_costs_stay_float32_for_a_bf16_cost_volume():
    """Under bf16 autocast the cost-volume head hands over a bf16 tensor; the trajectory costs must come back in float32
    (stp3/cost.py:36-47: a half cost volume indexed and added to float32 terms promotes) and equal the costs of the same
    values held in float32 -- not their bf16 rounding, which would re-rank near-tied samples."""
    from stp3_amd.cost import Cost_Function
    c = cfg()
    ins = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.planning_inputs(c).items()}
    cf = Cost_Function(c).cuda()
    lane, drv = split_hd(ins['hdmap_labels'])
    cv16 = ins['cost_volume'].to(torch.bfloat16)
    fc, fo = cf(cv16, ins['trajs'][..., :2].clone(), ins['occupancy'], lane.clone(), drv.clone(), ins['target'])
    fc32, fo32 = cf(cv16.float(), ins['trajs'][..., :2].clone(), ins['occupancy'], lane.clone(), drv.clone(), ins['target'])
    assert fc.dtype == torch.float32 and fo.dtype == torch.float32
    assert torch.equal(fc, fc32) and torch.equal(fo, fo32)
    assert not torch.equal(fo, fo.to(torch.bfloat16).float())            # the costs are not bf16 numbers


def test_cost_kernel_full_size_against_torch_statements():
    """nuscenes/Planning.yml sizes: 1 800 samples x 6 steps x batch 4.  The kernel against the module's own torch
    statements on float64 copies of the same inputs (they take the statement route), and bit-reproducible."""
    from stp3_amd.config import perception_cfg
    from stp3_amd.cost import Cost_Function
    from stp3_amd import synthetic
    c = perception_cfg(**{**PLANNING, 'N_FUTURE_FRAMES': 6, 'PLANNING.SAMPLE_NUM': 1800})
    B, T = 4, 6
    p = synthetic.make_planning_inputs(B, T, 1800, seed=5)
    trajs = p['sample_trajectory'][:, :, 1:, :2].cuda()
    g = torch.Generator().manual_seed(11)
    cv = (torch.randn(B, T, 200, 200, generator=g) * 0.5).cuda().requires_grad_(True)
    occ = (torch.rand(B, T, 200, 200, generator=g) < 0.02).cuda()
    hd = torch.randn(B, 4, 200, 200, generator=g).cuda()
    lane, drv = split_hd(hd)
    w = torch.randn(B, 1800, T, generator=g).cuda()
    cf = Cost_Function(c).cuda()
    fc, fo = cf(cv, trajs, occ, lane, drv, p['target_point'].cuda())
    (fo * w).sum().backward()
    cv64 = cv.detach().double().requires_grad_(True)
    fc64, fo64 = cf(cv64, trajs.double(), occ, lane.double(), drv.double(), p['target_point'].cuda().double())
    (fo64 * w.double()).sum().backward()
    assert fc64.dtype == torch.float64
    # the float64 statements see the same cells unless a coordinate lands within float32 rounding of a cell border
    close = torch.isclose(fo.double(), fo64, rtol=1e-4, atol=1e-4)
    assert close.float().mean().item() > 0.999, close.float().mean().item()
    assert torch.isclose(fc.double(), fc64, rtol=1e-4, atol=1e-3).float().mean().item() > 0.999
    assert torch.isclose(cv.grad.double(), cv64.grad, rtol=1e-4, atol=1e-4).float().mean().item() > 0.9999
    cv2 = cv.detach().clone().requires_grad_(True)
    fc2, fo2 = cf(cv2, trajs, occ, lane, drv, p['target_point'].cuda())
    (fo2 * w).sum().backward()
    assert torch.equal(fc, fc2) and torch.equal(fo, fo2) and torch.equal(cv.grad, cv2.grad)


def test_planning_config_training_and_validation_step():
    """A planning configuration end to end in bf16: 4 future frames, cost-volume head, planner loss in the total, a finite
    gradient on every trainable parameter; then a validation step that fills the planning metric."""
    from stp3_amd import synthetic
    from stp3_amd.trainer import TrainingModule
    from stp3_amd.utils import to_channels_last
    c = cfg()
    tm = to_channels_last(TrainingModule(c.convert_to_dict()).cuda())
    tm.train()
    batch = synthetic.make_batch(batch=1, seq=7, seed=3, planning=(c.N_FUTURE_FRAMES, c.PLANNING.SAMPLE_NUM))
    batch = {k: (v.cuda() if torch.is_tensor(v) and k not in ('intrinsics', 'extrinsics', 'future_egomotion') else v)
             for k, v in batch.items()}
    with torch.autocast('cuda', dtype=torch.bfloat16):
        output, labels, loss = tm.shared_step(batch, True)
    assert 'planning' in loss and output['selected_traj'].shape == (1, c.N_FUTURE_FRAMES + 1, 3)
    total = sum(loss.values())
    total.backward()
    assert torch.isfinite(total).item()
    bad = [n for n, p in tm.model.named_parameters() if p.requires_grad and (p.grad is None or not torch.isfinite(p.grad).all().item())]
    assert not bad, bad[:5]
    tm.eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        output, _, _ = tm.shared_step(batch, False)
    assert int(tm.metric_planning_val.total) == 1 and torch.isfinite(tm.metric_planning_val.compute()['L2']).all()
