"""How much of the float32 step's distance from the float64 truth is a draw of rounding noise?

    python scripts/step_noise_probe.py [--device cuda|cpu] [--draws 5] [--out profiles/r04_step_noise_probe.json]

Runs the product's float32 training step (BASELINE configs[2] at B = 2, top-k off: tests/test_step_parity_gpu.py's
``b2k0`` case against the reference's float64 fixture ``step_b2k0d.npz``) several times, each with the camera images
perturbed by a relative 1e-7 (one float32 ulp: a different but equally valid rounding of the same input), and once more
with the torch statements of the losses in place of the loss kernels (the state of the tree when
profiles/r03_parity_step.json was written).  If the spread over the draws covers the difference between two commits, that
difference is a draw of the step's rounding noise and not a change of arithmetic: the verdict of round 3 asked where
grad/temporal 0.027 -> 0.065 came from.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'st-p3_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--device', default='cuda')
    ap.add_argument('--draws', type=int, default=5)
    ap.add_argument('--scale', type=float, default=1e-7)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    os.environ['STP3_PARITY_DEVICE'] = a.device
    from stp3_amd import ops_loss
    from tests import test_step_parity_gpu as T
    rows = {}

    def one(tag):
        m, _ = T.measure('b2k0', fixture='b2k0d')
        keep = {k: v for k, v in m.items() if k.startswith('grad/') or k in ('tap_gout/temporal_model.final_conv',
                                                                            'tap_gout/decoder.layer1.0', 'tap_gout/decoder.up1_skip',
                                                                            'tap_out/temporal_model.final_conv')}
        rows[tag] = keep
        print(tag, ' '.join(f'{k.split("/", 1)[1]}={v:.3e}' for k, v in keep.items()), flush=True)

    one('unperturbed')
    for i in range(a.draws):
        T.PERTURB = (100 + i, a.scale)
        one(f'draw{i}')
    T.PERTURB = None
    real = ops_loss.supported
    ops_loss.supported = lambda x: False                       # the torch statements of the losses (stp3_amd/losses.py)
    try:
        one('torch_losses')
        T.PERTURB = (100, a.scale)
        one('torch_losses_draw0')
    finally:
        ops_loss.supported = real
        T.PERTURB = None
    noise = T.reference_noise()
    rows['reference_noise'] = {k: noise[k] for k in rows['unperturbed'] if k in noise}
    if a.out:
        json.dump(rows, open(a.out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
