"""Configuration tree with the reference's keys and defaults.

The reference builds its config on fvcore/yacs ``CfgNode`` (stp3/config.py:1-29, 164-189), which
is not installed here; ``CfgNode`` below is a small stand-in with the behaviour the path relies
on: attribute access, ``clone``, ``merge_from_file`` (YAML), ``merge_from_list`` (KEY VALUE ...),
``merge_from_other_cfg`` and ``convert_to_dict``.  Key names and default values are the
interface (stp3/config.py:32-162); reference YAML files load unchanged.
"""
import argparse
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def convert_to_dict(self):
        return {k: (v.convert_to_dict() if isinstance(v, CfgNode) else copy.deepcopy(v)) for k, v in self.items()}

    # ---- merging -------------------------------------------------------------------------
    def _merge(self, other, path):
        for k, v in other.items():
            if k not in self:
                raise KeyError('Non-existent config key: {}'.format('.'.join(path + [k])))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError('{} must be a mapping'.format('.'.join(path + [k])))
                self[k]._merge(v, path + [k])
            else:
                self[k] = _coerce(v, self[k], '.'.join(path + [k]))

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_file(self, filename):
        with open(filename) as f:
            self._merge(yaml.safe_load(f) or {}, [])

    def merge_from_list(self, opts):
        opts = list(opts or [])
        if len(opts) % 2:
            raise ValueError('override list must be KEY VALUE pairs')
        for key, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f'Non-existent config key: {key}')
            if isinstance(raw, str):
                try:
                    raw = ast.literal_eval(raw)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = _coerce(raw, node[parts[-1]], key)


def _coerce(value, current, key):
    """yacs-style type agreement: tuples/lists interchange, ints promote to float."""
    if isinstance(current, float) and isinstance(value, int) and not isinstance(value, bool):
        return float(value)
    if isinstance(current, float) and isinstance(value, str):
        return float(value)          # YAML reads `1e-3` as a string
    if isinstance(current, tuple) and isinstance(value, list):
        return tuple(value)
    if isinstance(current, list) and isinstance(value, tuple):
        return list(value)
    if current is not None and value is not None and type(current) is not type(value) \
            and not (isinstance(current, (list, tuple)) and isinstance(value, (list, tuple))):
        raise ValueError(f'type mismatch for {key}: {type(current).__name__} vs {type(value).__name__}')
    return value


def _defaults():
    c = CfgNode()
    c.LOG_DIR = 'tensorboard_logs'
    c.TAG = 'default'
    c.GPUS = [0]
    c.PRECISION = 32
    c.BATCHSIZE = 3
    c.EPOCHS = 20
    c.N_WORKERS = 5
    c.VIS_INTERVAL = 5000
    c.LOGGING_INTERVAL = 500
    c.PRETRAINED = CfgNode(dict(LOAD_WEIGHTS=False, PATH=''))
    c.DATASET = CfgNode(dict(DATAROOT='/data/Nuscenes', VERSION='trainval', NAME='nuscenes',
                             MAP_FOLDER='/data/Nuscenes', IGNORE_INDEX=255, FILTER_INVISIBLE_VEHICLES=True,
                             SAVE_DIR='datas'))
    c.TIME_RECEPTIVE_FIELD = 3
    c.N_FUTURE_FRAMES = 4
    c.IMAGE = CfgNode(dict(FINAL_DIM=(224, 480), RESIZE_SCALE=0.3, TOP_CROP=46, ORIGINAL_HEIGHT=900,
                           ORIGINAL_WIDTH=1600,
                           NAMES=['CAM_FRONT_LEFT', 'CAM_FRONT', 'CAM_FRONT_RIGHT', 'CAM_BACK_LEFT', 'CAM_BACK',
                                  'CAM_BACK_RIGHT']))
    c.LIFT = CfgNode(dict(X_BOUND=[-50.0, 50.0, 0.5], Y_BOUND=[-50.0, 50.0, 0.5], Z_BOUND=[-10.0, 10.0, 20.0],
                          D_BOUND=[2.0, 50.0, 1.0], GT_DEPTH=False, DISCOUNT=0.5))
    c.EGO = CfgNode(dict(WIDTH=1.85, HEIGHT=4.084))
    c.MODEL = CfgNode(dict(
        ENCODER=dict(DOWNSAMPLE=8, NAME='efficientnet-b4', OUT_CHANNELS=64, USE_DEPTH_DISTRIBUTION=True),
        TEMPORAL_MODEL=dict(NAME='temporal_block', START_OUT_CHANNELS=64, EXTRA_IN_CHANNELS=0, INBETWEEN_LAYERS=0,
                            PYRAMID_POOLING=True, INPUT_EGOPOSE=True),
        DISTRIBUTION=dict(LATENT_DIM=32, MIN_LOG_SIGMA=-5.0, MAX_LOG_SIGMA=5.0),
        FUTURE_PRED=dict(N_GRU_BLOCKS=2, N_RES_LAYERS=1, MIXTURE=True),
        DECODER=dict(),
        BN_MOMENTUM=0.1))
    c.SEMANTIC_SEG = CfgNode(dict(
        VEHICLE=dict(WEIGHTS=[1.0, 2.0], USE_TOP_K=True, TOP_K_RATIO=0.25),
        PEDESTRIAN=dict(ENABLED=True, WEIGHTS=[1.0, 10.0], USE_TOP_K=True, TOP_K_RATIO=0.25),
        HDMAP=dict(ENABLED=True, ELEMENTS=['lane_divider', 'drivable_area'], WEIGHTS=[[1.0, 5.0], [1.0, 1.0]],
                   TRAIN_WEIGHT=[1, 1], USE_TOP_K=[True, False], TOP_K_RATIO=[0.25, 0.25])))
    c.INSTANCE_SEG = CfgNode(dict(ENABLED=True))
    c.INSTANCE_FLOW = CfgNode(dict(ENABLED=True))
    c.PROBABILISTIC = CfgNode(dict(ENABLED=True, METHOD='GAUSSIAN'))
    c.PLANNING = CfgNode(dict(ENABLED=True, GRU_STATE_SIZE=256, SAMPLE_NUM=600, COMMAND=['LEFT', 'FORWARD', 'RIGHT']))
    c.FUTURE_DISCOUNT = 0.95
    c.OPTIMIZER = CfgNode(dict(LR=3e-4, WEIGHT_DECAY=1e-7))
    c.GRAD_NORM_CLIP = 5
    c.COST_FUNCTION = CfgNode(dict(SAFETY=0.1, LAMBDA=1.0, HEADWAY=1.0, LRDIVIDER=10.0, COMFORT=0.1, PROGRESS=0.5,
                                   VOLUME=100.0))
    return c


_C = _defaults()

# stp3/configs/nuscenes/Perception.yml -- the scripts/train_perceive.sh configuration this path is
# benchmarked on (BASELINE.json configs[1..3]); kept here so bench/tests need no YAML on disk.
PERCEPTION_OVERRIDES = {
    'TAG': 'Perception', 'GPUS': [0, 1, 2, 3], 'BATCHSIZE': 3, 'PRECISION': 16, 'EPOCHS': 20, 'N_WORKERS': 8,
    'DATASET': {'VERSION': 'trainval'}, 'TIME_RECEPTIVE_FIELD': 3, 'N_FUTURE_FRAMES': 0,
    'LIFT': {'GT_DEPTH': False},
    'MODEL': {'ENCODER': {'NAME': 'efficientnet-b4', 'USE_DEPTH_DISTRIBUTION': True},
              'TEMPORAL_MODEL': {'NAME': 'temporal_block', 'INPUT_EGOPOSE': True}, 'BN_MOMENTUM': 0.05},
    'SEMANTIC_SEG': {'PEDESTRIAN': {'ENABLED': True}, 'HDMAP': {'ENABLED': True}},
    'INSTANCE_SEG': {'ENABLED': False}, 'INSTANCE_FLOW': {'ENABLED': False},
    'PROBABILISTIC': {'ENABLED': False}, 'PLANNING': {'ENABLED': False}, 'OPTIMIZER': {'LR': 1e-3},
}


def get_parser():
    """Same command line as the reference (stp3/config.py:164-170)."""
    parser = argparse.ArgumentParser(description='ST-P3 (MI355X) training')
    parser.add_argument('--config-file', default='', metavar='FILE', help='path to config file')
    parser.add_argument('opts', help='Modify config options using the command-line', default=None,
                        nargs=argparse.REMAINDER)
    return parser


def get_cfg(args=None, cfg_dict=None):
    """Defaults <- cfg_dict <- YAML file <- KEY VALUE overrides (stp3/config.py:173-189)."""
    cfg = _C.clone()
    if cfg_dict is not None:
        cfg.merge_from_other_cfg(CfgNode(cfg_dict))
    if args is not None:
        if args.config_file:
            cfg.merge_from_file(args.config_file)
        cfg.merge_from_list(args.opts)
    return cfg


def perception_cfg(**overrides):
    """Perception.yml (+ dotted-key overrides, e.g. ``perception_cfg(**{'LIFT.GT_DEPTH': True})``)."""
    cfg = get_cfg(cfg_dict=PERCEPTION_OVERRIDES)
    flat = []
    for k, v in overrides.items():
        flat += [k, v]
    cfg.merge_from_list(flat)
    return cfg
