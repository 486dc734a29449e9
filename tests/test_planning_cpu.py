"""CPU: the planner against fixtures from the reference's own classes
(oracle/make_golden_planning.py -> tests/golden/planning.npz): the ego footprint tables, every term of Cost_Function for
label and logit hd maps, the cost-volume gradient, Planning (loss, refined trajectory, gradients; eval trajectory),
PlanningMetric, and the state-dict keys of the reference's TrainingModule with the planner enabled.  CPU tensors take
the torch statements of stp3_amd/cost.py; the HIP kernel behind the same calls is checked on the CPU stand-in
(tests/test_kernels_on_cpu.py, case `plan`) and on the MI355X (tests/test_planning_gpu.py)."""
import json
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

PLANNING = {'N_FUTURE_FRAMES': 4, 'PLANNING.ENABLED': True, 'PLANNING.SAMPLE_NUM': 60, 'PROBABILISTIC.ENABLED': False,
            'SEMANTIC_SEG.PEDESTRIAN.ENABLED': True, 'SEMANTIC_SEG.HDMAP.ENABLED': True, 'INSTANCE_FLOW.ENABLED': False,
            'INSTANCE_SEG.ENABLED': False}
# float32 costs: the kernel and the reference differ by single roundings (torch's CPU square root is not correctly
# rounded, the library's is); costs are O(1..100)
COST_TOL = dict(rtol=1e-5, atol=2e-5)


def cfg():
    from stp3_amd.config import perception_cfg
    return perception_cfg(**PLANNING)


def split_hd(hd):
    return (hd[:, 0:1], hd[:, 1:2]) if hd.shape[1] == 2 else (hd[:, 0:2], hd[:, 2:4])


def cost_case(form, device='cpu'):
    """(cost_fc, cost_fo, d cost_volume, cost_fc without a target) of Cost_Function on the fixture inputs."""
    from stp3_amd.cost import Cost_Function
    c = cfg()
    ins = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in H.planning_inputs(c).items()}
    cf = Cost_Function(c).to(device)
    lane, drv = split_hd(ins['hdmap_labels'] if form == 'train' else ins['hdmap_logits'])
    cv = ins['cost_volume'].clone().requires_grad_(True)
    fc, fo = cf(cv, ins['trajs'][..., :2].clone(), ins['occupancy'], lane.clone(), drv.clone(), ins['target'])
    (fo * ins['w_fo']).sum().backward()
    fc0, _ = cf(cv.detach(), ins['trajs'][..., :2].clone(), ins['occupancy'], lane.clone(), drv.clone(),
                torch.zeros_like(ins['target']))
    return fc.detach().cpu(), fo.detach().cpu(), cv.grad.cpu(), fc0.cpu()


def check_costs(form, device):
    g = H.load('planning.npz')
    fc, fo, dcv, fc0 = cost_case(form, device)
    np.testing.assert_allclose(fc.numpy(), g[f'{form}/cost_fc'], **COST_TOL)
    np.testing.assert_allclose(fo.numpy(), g[f'{form}/cost_fo'], **COST_TOL)
    np.testing.assert_allclose(fc0.numpy(), g[f'{form}/cost_fc_no_target'], **COST_TOL)
    # the gradient is a scatter of the weights through the clamps: exact up to the clamp decisions
    np.testing.assert_allclose(dcv.numpy(), g[f'{form}/d_cost_volume'], rtol=1e-6, atol=1e-6)


def planner_case(device='cpu', autocast=False):
    from stp3_amd.models.planning_model import Planning
    from tests.test_train_parity_gpu import make_deterministic_train
    c = cfg()
    ins = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in H.planning_inputs(c).items()}
    pl = Planning(c, 64, 6, gru_state_size=c.PLANNING.GRU_STATE_SIZE)
    for sub in (pl.reduce_channel, pl.GRU, pl.decoder):
        H.fill_deterministic(sub)
    pl = make_deterministic_train(pl).to(device)
    cv = ins['cost_volume'].clone().requires_grad_(True)
    cam = ins['cam_front'].clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        loss, traj = pl(cam, ins['sample_trajs'].clone(), ins['gt_trajs'].clone(), cv, ins['occupancy'], ins['hdmap_labels'],
                        ins['commands'], ins['target'])
    loss.float().backward()
    out = {'loss': loss.item(), 'traj': traj.detach().float().cpu(), 'd_cost_volume': cv.grad.cpu(),
           'd_cam_front': H.sample(cam.grad).cpu(), 'gnorm': {n: p.grad.double().norm().item() for n, p in pl.named_parameters()
                                                               if p.grad is not None}}
    pl.eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
        loss_e, traj_e = pl(ins['cam_front'], ins['sample_trajs'].clone(), ins['gt_trajs'].clone(), ins['cost_volume'],
                            ins['occupancy'], ins['hdmap_logits'], ins['commands'], ins['target'])
    assert loss_e == 0
    out['eval_traj'] = traj_e.float().cpu()
    return out


def check_planner(out, tol, gtol):
    g = H.load('planning.npz')
    assert abs(out['loss'] - g['planner/train/loss'][0]) <= tol * abs(g['planner/train/loss'][0])
    np.testing.assert_allclose(out['traj'].numpy(), g['planner/train/traj'], rtol=tol, atol=tol)
    np.testing.assert_allclose(out['eval_traj'].numpy(), g['planner/eval/traj'], rtol=tol, atol=tol)
    np.testing.assert_allclose(out['d_cost_volume'].numpy(), g['planner/train/d_cost_volume'], rtol=1e-5, atol=1e-5)
    ref = g['planner/train/d_cam_front']
    assert np.abs(out['d_cam_front'].numpy() - ref).max() <= gtol * np.abs(ref).max()
    for name, norm in out['gnorm'].items():
        want = g[f'planner/train/gnorm/{name}'][0]
        if want > 1e-4:                                    # (biases in front of a BatchNorm have noise-level gradients)
            assert abs(norm - want) <= gtol * want, (name, norm, want)


def test_footprints_match_the_reference():
    from stp3_amd.cost import BaseCost
    g = H.load('planning.npz')
    base = BaseCost(cfg())
    assert np.array_equal(base.footprint(0), g['footprint0']) and len(g['footprint0']) == 32      # metrics.py:313
    assert np.array_equal(base.footprint(2), g['footprint_lambda'])


def test_footprint_with_lattice_points_on_the_outline():
    """Box edges on cell boundaries (e.g. EGO.WIDTH = 2.0 at 0.5 m cells): decided by the half-open crossing test of the
    scikit-image release the reference pins (0.18.1) -- first row / column edge in, last one out -- and equal to the
    independent restatement the oracle runs the reference with; nothing raises."""
    from oracle.ref_stubs import polygon
    from stp3_amd.cost import BaseCost, footprint_cells
    box = np.array([[1.0, 1.5], [4.0, 1.5], [4.0, 3.5], [1.0, 3.5]])                   # rows 1 and 4 lie on edges
    cells = footprint_cells(box)
    assert cells.tolist() == [[r, c] for r in (1, 2, 3) for c in (2, 3)]
    rr, cc = polygon(box[:, 0], box[:, 1])
    assert np.array_equal(cells, np.stack([rr, cc], axis=-1))
    both = np.array([[1.0, 1.0], [4.0, 1.0], [4.0, 3.0], [1.0, 3.0]])                  # all four edges on lattice lines
    rr, cc = polygon(both[:, 0], both[:, 1])
    assert np.array_equal(footprint_cells(both), np.stack([rr, cc], axis=-1))
    assert footprint_cells(both).tolist() == [[r, c] for r in (1, 2, 3) for c in (1, 2)]
    cells = footprint_cells(np.array([[0.5, 1.5], [4.5, 1.5], [4.5, 3.5], [0.5, 3.5]]))
    assert cells.tolist() == [[r, c] for r in (1, 2, 3, 4) for c in (2, 3)]
    # a configuration whose ego box is 2 m x 4 m on the 0.5 m grid builds its tables (it used to raise)
    from stp3_amd.config import perception_cfg
    c = perception_cfg(**{**PLANNING, 'EGO.WIDTH': 2.0, 'EGO.HEIGHT': 4.0})
    base = BaseCost(c)
    rr, cc = polygon(*(((np.array([[-1.5, 1.0], [2.5, 1.0], [2.5, -1.0], [-1.5, -1.0]]) - base.bx.numpy()) / base.dx.numpy()).T))
    assert np.array_equal(base.footprint(0), np.stack([rr, cc], axis=-1)) and len(rr) == 32


def test_kernel_tables_follow_a_load_state_dict():
    """Cost_Function caches the footprint tables / scalar parameters of the kernel path: a ``load_state_dict`` that
    changes safetycost.w / dx / bx after the first use must rebuild them."""
    from stp3_amd.cost import Cost_Function
    cf = Cost_Function(cfg())
    fp0, fpl, params = cf._kernel_inputs('cpu')
    again = cf._kernel_inputs('cpu')
    assert again[0] is fp0 and again[2] is params                                      # cached while nothing changes
    sd = cf.state_dict()
    sd['safetycost.w'] = sd['safetycost.w'] * 2.0
    sd['safetycost.dx'] = sd['safetycost.dx'] * 2.0                                    # 1 m cells: a smaller footprint
    sd['headwaycost.dx'] = sd['headwaycost.dx'] * 2.0
    cf.load_state_dict(sd)
    fp0b, fplb, paramsb = cf._kernel_inputs('cpu')
    assert float(paramsb['w0']) == 2.0 * float(params['w0']) and float(paramsb['dx0']) == 2.0 * float(params['dx0'])
    assert fp0b.shape[0] < fp0.shape[0] and len(cf._tables) == 1
    assert np.array_equal(cf.safetycost.footprint(0), fp0b.numpy())


@pytest.mark.parametrize('form', ['train', 'eval'])
def test_cost_terms_match_the_reference(form):
    """Each of the seven terms separately (the fused kernel only returns their sums)."""
    from stp3_amd.cost import Cost_Function
    c = cfg()
    g = H.load('planning.npz')
    ins = H.planning_inputs(c)
    cf = Cost_Function(c)
    lane, drv = split_hd(ins['hdmap_labels'] if form == 'train' else ins['hdmap_logits'])
    tr = ins['trajs'][..., :2] * torch.tensor([-1.0, 1.0])
    occ = ins['occupancy'].float()
    terms = {'safety': cf.safetycost(tr, occ), 'headway': cf.headwaycost(tr, occ, drv), 'lrdivider': cf.lrdividercost(tr, lane),
             'comfort': cf.comfortcost(tr), 'progress': cf.progresscost(tr, ins['target']), 'rule': cf.rulecost(tr, drv),
             'volume': cf.costvolume(tr, ins['cost_volume'])}
    for name, v in terms.items():
        np.testing.assert_allclose(v.float().numpy(), g[f'{form}/term/{name}'], rtol=1e-6, atol=1e-6, err_msg=name)


@pytest.mark.parametrize('form', ['train', 'eval'])
def test_cost_function_matches_the_reference(form):
    check_costs(form, 'cpu')


def test_planner_matches_the_reference():
    check_planner(planner_case(), tol=1e-5, gtol=1e-4)


def test_planning_metric_matches_the_reference():
    from stp3_amd.metrics import PlanningMetric
    c = cfg()
    g = H.load('planning.npz')
    ins = H.planning_inputs(c)
    m = PlanningMetric(c, c.N_FUTURE_FRAMES)
    m(ins['sample_trajs'][:, 7].clone(), ins['gt_trajs'].clone(), ins['occupancy'])
    m(*H.planning_metric_trajs(c), ins['occupancy'])
    for k in ('obj_col', 'obj_box_col', 'L2', 'total'):
        np.testing.assert_allclose(getattr(m, k).numpy(), g[f'metric/{k}'], rtol=1e-6)
    assert g['metric/obj_col'].sum() > 0 and g['metric/obj_box_col'].sum() > 0          # the fixture has collisions
    scores = m.compute()
    assert scores['L2'].shape == (c.N_FUTURE_FRAMES,)


def test_planning_config_state_dict_matches_reference():
    from stp3_amd.trainer import TrainingModule
    want = json.load(open(os.path.join(H.GOLDEN, 'state_dict_keys.json')))['TrainingModule_planning']
    got = {k: list(v.shape) for k, v in TrainingModule(cfg().convert_to_dict()).state_dict().items()}
    missing, extra = sorted(set(want) - set(got)), sorted(set(got) - set(want))
    assert not missing and not extra, (missing[:5], extra[:5])
    assert not [k for k in want if want[k] != got[k]]
    assert any(k.startswith('model.planning.cost_function.safetycost.') for k in want) and 'model.planning_weight' in want
