"""TEST INFRASTRUCTURE (build container only) -- fixtures of the planner from the
REFERENCE's own classes.

    python oracle/make_golden_planning.py    -> tests/golden/planning.npz, state_dict_keys.json['TrainingModule_planning']

Float32 CPU.  ``skimage.draw.polygon`` (absent here) is the restatement in oracle/ref_stubs.py.
  * ``Cost_Function`` and each of its seven terms                         stp3/cost.py:10-392
      - training form: occupancy = bool labels, hd map = (B,2,H,W) integer labels
      - evaluation form: hd map = (B,4,H,W) logits (softmax + threshold inside the terms)
    costs, and the gradient of a weighted sum of them with respect to the cost volume
  * ``Planning`` (train(): loss + refined trajectory + gradients; eval(): trajectory)   stp3/models/planning_model.py:10-150
  * ``PlanningMetric`` (L2, box collisions)                                stp3/metrics.py:263-372
  * parameter / buffer names of the reference ``TrainingModule`` for a planning config
Inputs are exact-integer pseudo-random (tests/helpers.det_tensor) so the GPU box rebuilds them bit for bit.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))

from oracle import ref_stubs  # noqa: E402
from oracle.make_golden_train import grad_samples, install_trainer_stubs, make_deterministic_train  # noqa: E402
from stp3_amd.config import perception_cfg  # noqa: E402
from stp3_amd.models.efficientnet import EfficientNet as OurEfficientNet  # noqa: E402
from stp3_amd.models.resnet import resnet18 as our_resnet18  # noqa: E402
from tests import helpers as H  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
PLANNING = {'N_FUTURE_FRAMES': 4, 'PLANNING.ENABLED': True, 'PLANNING.SAMPLE_NUM': 60, 'PROBABILISTIC.ENABLED': False,
            'SEMANTIC_SEG.PEDESTRIAN.ENABLED': True, 'SEMANTIC_SEG.HDMAP.ENABLED': True, 'INSTANCE_FLOW.ENABLED': False,
            'INSTANCE_SEG.ENABLED': False}


def main():
    torch.manual_seed(0)
    ref_stubs.install(efficientnet_cls=OurEfficientNet, resnet18_fn=our_resnet18)
    install_trainer_stubs()
    from stp3.cost import Cost_Function
    from stp3.metrics import PlanningMetric
    from stp3.models.planning_model import Planning
    from stp3.trainer import TrainingModule
    cfg = perception_cfg(**PLANNING)
    out = {}
    ins = H.planning_inputs(cfg)                          # the same builder the product's tests call
    cf = Cost_Function(cfg)
    out['footprint0'] = cf.safetycost.get_origin_points(0).numpy()
    out['footprint_lambda'] = cf.safetycost.get_origin_points(int(cfg.COST_FUNCTION.LAMBDA / 0.5)).numpy()

    for form in ('train', 'eval'):
        hd = ins['hdmap_labels'] if form == 'train' else ins['hdmap_logits']
        lane, drv = (hd[:, 0:1], hd[:, 1:2]) if hd.shape[1] == 2 else (hd[:, 0:2], hd[:, 2:4])
        cv = ins['cost_volume'].clone().requires_grad_(True)
        flipped = ins['trajs'][..., :2] * torch.tensor([-1, 1])
        terms = {'safety': cf.safetycost(flipped.clone(), ins['occupancy']),
                 'headway': cf.headwaycost(flipped.clone(), ins['occupancy'], drv.clone()),
                 'lrdivider': cf.lrdividercost(flipped.clone(), lane.clone()),
                 'comfort': cf.comfortcost(flipped.clone()),
                 'progress': cf.progresscost(flipped.clone(), ins['target']),
                 'rule': cf.rulecost(flipped.clone(), drv.clone()),
                 'volume': cf.costvolume(flipped.clone(), cv.detach())}
        for k, v in terms.items():
            out[f'{form}/term/{k}'] = v.detach().float().numpy()
        fc, fo = cf(cv, ins['trajs'][..., :2].clone(), ins['occupancy'], lane.clone(), drv.clone(), ins['target'])
        out[f'{form}/cost_fc'], out[f'{form}/cost_fo'] = fc.detach().numpy(), fo.detach().numpy()
        (fo * ins['w_fo']).sum().backward()
        out[f'{form}/d_cost_volume'] = cv.grad.numpy()
        # no target point (the reference tests the SUM of the batch's target points)
        fc0, _ = cf(cv.detach(), ins['trajs'][..., :2].clone(), ins['occupancy'], lane.clone(), drv.clone(),
                    torch.zeros_like(ins['target']))
        out[f'{form}/cost_fc_no_target'] = fc0.numpy()

    planner = Planning(cfg, 64, 6, gru_state_size=cfg.PLANNING.GRU_STATE_SIZE)
    for sub in (planner.reduce_channel, planner.GRU, planner.decoder):
        H.fill_deterministic(sub)
    make_deterministic_train(planner)
    cv = ins['cost_volume'].clone().requires_grad_(True)
    cam = ins['cam_front'].clone().requires_grad_(True)
    loss, traj = planner(cam, ins['sample_trajs'].clone(), ins['gt_trajs'].clone(), cv, ins['occupancy'],
                         ins['hdmap_labels'], ins['commands'], ins['target'])
    loss.backward()
    out['planner/train/loss'] = np.array([loss.item()])
    out['planner/train/traj'] = traj.detach().numpy()
    out['planner/train/d_cost_volume'] = cv.grad.numpy()
    out['planner/train/d_cam_front'] = H.sample(cam.grad).numpy()
    grad_samples(planner, 'planner/train', out)
    planner.eval()
    with torch.no_grad():
        loss_e, traj_e = planner(ins['cam_front'], ins['sample_trajs'].clone(), ins['gt_trajs'].clone(), ins['cost_volume'],
                                 ins['occupancy'], ins['hdmap_logits'], ins['commands'], ins['target'])
    out['planner/eval/traj'] = traj_e.numpy()
    assert loss_e == 0

    metric = PlanningMetric(cfg, cfg.N_FUTURE_FRAMES)
    metric.update(ins['sample_trajs'][:, 7].clone(), ins['gt_trajs'].clone(), ins['occupancy'])
    plan, expert = H.planning_metric_trajs(cfg)           # a plan that drives through the obstacle, an expert beside it
    metric.update(plan.clone(), expert.clone(), ins['occupancy'])
    for k in ('obj_col', 'obj_box_col', 'L2', 'total'):
        out[f'metric/{k}'] = getattr(metric, k).numpy()

    np.savez_compressed(os.path.join(GOLDEN, 'planning.npz'), **out)
    ref = TrainingModule(cfg.convert_to_dict())
    keys = {k: list(v.shape) for k, v in ref.state_dict().items()}
    path = os.path.join(GOLDEN, 'state_dict_keys.json')
    allkeys = json.load(open(path))
    allkeys['TrainingModule_planning'] = keys
    json.dump(allkeys, open(path, 'w'), indent=0, sort_keys=True)
    man_path = os.path.join(GOLDEN, 'MANIFEST.json')
    man = json.load(open(man_path))
    man['planning'] = {'file': 'planning.npz', 'generator': 'oracle/make_golden_planning.py', 'entries': len(out),
                       'what': 'reference Cost_Function (all seven terms, training and evaluation forms, cost-volume '
                               'gradient), Planning (loss, refined trajectory, gradients; eval trajectory), PlanningMetric; '
                               'float32 CPU; skimage.draw.polygon restated (oracle/ref_stubs.py); + state-dict keys of a '
                               'planning config'}
    json.dump(man, open(man_path, 'w'), indent=1, sort_keys=True)
    print(len(out), 'arrays;', len(keys), 'state-dict keys (planning config)')
    print('footprints', out['footprint0'].shape, out['footprint_lambda'].shape)
    for form in ('train', 'eval'):
        print(form, {k[len(form) + 6:]: float(np.abs(v).mean()) for k, v in out.items() if k.startswith(form + '/term/')})
    print('loss', out['planner/train/loss'], 'traj', out['planner/train/traj'][0, :2])


if __name__ == '__main__':
    main()
