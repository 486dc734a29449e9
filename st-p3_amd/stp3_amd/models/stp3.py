"""``STP3`` -- drop-in for the reference's ``stp3.models.stp3.STP3`` on the perception path.

Same constructor (``STP3(cfg)``), attributes (``receptive_field, n_future, encoder_downsample,
bev_dimension, encoder, temporal_model, decoder`` and the non-trainable parameters
``bev_resolution / bev_start_position / bev_dimension / frustum``, stp3/models/stp3.py:15-109) and
``forward(image, intrinsics, extrinsics, future_egomotion) -> dict`` (:132-184).  What differs is
how the middle is computed: ``get_geometry`` + ``encoder_forward``'s outer product +
``projection_to_birds_eye_view`` (:186-301) are replaced by the HIP operators in ``stp3_amd.ops``
(one index/plan pass on a side stream that overlaps the image encoder, then a fused softmax /
voxel-pool / discounted-accumulate kernel), so the 1.5 GB lifted tensor and the per-(b,t) host
syncs of the reference never exist.

The prediction stage (``N_FUTURE_FRAMES > 0``: ``present_distribution``, ``future_prediction``, stp3.py:69-87, 157-176)
and the planner (``PLANNING.ENABLED``: ``planning``, stp3.py:104-107) are wired in as in
the reference; ``forward`` then also returns the present frame's front-camera features (``cam_front``) for the planner.
"""
import torch
import torch.nn as nn

from .. import ops
from .decoder import Decoder
from .distributions import DistributionModule
from .encoder import Encoder
from .future_prediction import FuturePrediction
from .planning_model import Planning
from .temporal_model import TemporalModel, TemporalModelIdentity


def bev_parameters(x_bounds, y_bounds, z_bounds):
    """Resolution, first-cell centre and size of the BEV grid (stp3/utils/geometry.py:40-59)."""
    rows = (x_bounds, y_bounds, z_bounds)
    resolution = torch.tensor([r[2] for r in rows])
    start = torch.tensor([r[0] + r[2] / 2.0 for r in rows])
    dimension = torch.tensor([(r[1] - r[0]) / r[2] for r in rows], dtype=torch.long)
    return resolution, start, dimension


def set_bn_momentum(model, momentum=0.1):
    """stp3/utils/network.py:27-30 -- applied to every BatchNorm, backbone included."""
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = momentum


class STP3(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        res, start, dim = bev_parameters(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
        self.bev_resolution = nn.Parameter(res, requires_grad=False)
        self.bev_start_position = nn.Parameter(start, requires_grad=False)
        self.bev_dimension = nn.Parameter(dim, requires_grad=False)

        self.encoder_downsample = cfg.MODEL.ENCODER.DOWNSAMPLE
        self.encoder_out_channels = cfg.MODEL.ENCODER.OUT_CHANNELS
        self.frustum = self.create_frustum()
        self.depth_channels = self.frustum.shape[0]
        self.discount = cfg.LIFT.DISCOUNT

        if cfg.TIME_RECEPTIVE_FIELD == 1:
            assert cfg.MODEL.TEMPORAL_MODEL.NAME == 'identity'
        self.receptive_field = cfg.TIME_RECEPTIVE_FIELD
        self.n_future = cfg.N_FUTURE_FRAMES
        self.latent_dim = cfg.MODEL.DISTRIBUTION.LATENT_DIM
        self.spatial_extent = (cfg.LIFT.X_BOUND[1], cfg.LIFT.Y_BOUND[1])
        self.bev_size = (int(dim[0]), int(dim[1]))
        if not cfg.MODEL.ENCODER.USE_DEPTH_DISTRIBUTION:
            # the reference's encoder returns depth = None for this setting and its own encoder_forward then calls
            # ``depth.view`` on it (stp3/models/stp3.py:217-222): the variant raises AttributeError in the reference before a
            # BEV tensor exists, so there is no behaviour to reproduce (tests/test_oracle_golden.py pins that)
            raise NotImplementedError('MODEL.ENCODER.USE_DEPTH_DISTRIBUTION=False is undefined in the reference '
                                      '(stp3/models/stp3.py:222 dereferences the missing depth tensor)')

        self.encoder = Encoder(cfg=cfg.MODEL.ENCODER, D=self.depth_channels)

        temporal_in = self.encoder_out_channels + (6 if cfg.MODEL.TEMPORAL_MODEL.INPUT_EGOPOSE else 0)
        name = cfg.MODEL.TEMPORAL_MODEL.NAME
        if name == 'identity':
            self.temporal_model = TemporalModelIdentity(temporal_in, self.receptive_field)
        elif name == 'temporal_block':
            tm = cfg.MODEL.TEMPORAL_MODEL
            self.temporal_model = TemporalModel(
                temporal_in, self.receptive_field, input_shape=self.bev_size,
                start_out_channels=tm.START_OUT_CHANNELS, extra_in_channels=tm.EXTRA_IN_CHANNELS,
                n_spatial_layers_between_temporal_layers=tm.INBETWEEN_LAYERS,
                use_pyramid_pooling=tm.PYRAMID_POOLING)
        else:
            raise NotImplementedError(f'Temporal module {name}.')
        self.future_pred_in_channels = self.temporal_model.out_channels
        if self.n_future > 0:
            # prediction stage (stp3.py:69-87): present distribution + future prediction
            if cfg.PROBABILISTIC.ENABLED:
                self.present_distribution = DistributionModule(self.future_pred_in_channels, self.latent_dim,
                                                               method=cfg.PROBABILISTIC.METHOD)
            fp = cfg.MODEL.FUTURE_PRED
            self.future_prediction = FuturePrediction(in_channels=self.future_pred_in_channels, latent_dim=self.latent_dim,
                                                      n_future=self.n_future, mixture=fp.MIXTURE,
                                                      n_gru_blocks=fp.N_GRU_BLOCKS, n_res_layers=fp.N_RES_LAYERS)

        self.decoder = Decoder(
            in_channels=self.future_pred_in_channels,
            n_classes=len(cfg.SEMANTIC_SEG.VEHICLE.WEIGHTS),
            n_present=self.receptive_field,
            n_hdmap=len(cfg.SEMANTIC_SEG.HDMAP.ELEMENTS),
            predict_gate={'perceive_hdmap': cfg.SEMANTIC_SEG.HDMAP.ENABLED,
                          'predict_pedestrian': cfg.SEMANTIC_SEG.PEDESTRIAN.ENABLED,
                          'predict_instance': cfg.INSTANCE_SEG.ENABLED,
                          'predict_future_flow': cfg.INSTANCE_FLOW.ENABLED,
                          'planning': cfg.PLANNING.ENABLED})
        if cfg.PLANNING.ENABLED:
            # stp3.py:104-107: trajectory costs + GRU refinement (models/planning_model.py, csrc/stp3_plan.hip)
            self.planning = Planning(cfg, self.encoder_out_channels, 6, gru_state_size=cfg.PLANNING.GRU_STATE_SIZE)
        set_bn_momentum(self, cfg.MODEL.BN_MOMENTUM)

        # BEV features in channels-last memory (same logical (B,T,C,X,Y) tensor): what the NHWC convolutions of the
        # temporal model read; False gives the reference's contiguous layout (stp3.py:230-232)
        self.bev_channels_last = True
        self._grid = None
        self._side_stream = None
        self.prebuilt_plan = None           # set by ``prepare_plan``: forward then does no host work at all

    # ------------------------------------------------------------------------------------------
    def create_frustum(self):
        """(D, fH, fW, 3) image-plane sample grid (x_px, y_px, depth), stp3.py:111-130."""
        h, w = self.cfg.IMAGE.FINAL_DIM
        fh, fw = h // self.encoder_downsample, w // self.encoder_downsample
        depth = torch.arange(*self.cfg.LIFT.D_BOUND, dtype=torch.float)
        n_d = depth.shape[0]
        grid = torch.stack((torch.linspace(0, w - 1, fw, dtype=torch.float).view(1, 1, fw).expand(n_d, fh, fw),
                            torch.linspace(0, h - 1, fh, dtype=torch.float).view(1, fh, 1).expand(n_d, fh, fw),
                            depth.view(n_d, 1, 1).expand(n_d, fh, fw)), -1)
        return nn.Parameter(grid, requires_grad=False)

    def lift_grid(self, device):
        """Device constants of the lift (frustum axes, BEV grid); rebuilt if the parameters moved."""
        if self._grid is None or self._grid.device != torch.device(device):
            self._grid = ops.LiftGrid(self.frustum, self.bev_resolution, self.bev_start_position,
                                      self.bev_dimension, device)
        return self._grid

    def prepare_plan(self, intrinsics, extrinsics, future_egomotion, device, out=None):
        """Build the geometry-only pooling plan of a batch ahead of ``forward`` (host-side pose math, one
        small upload, index kernels).  With a plan prepared, ``forward`` touches no host data, so the
        whole training step can be captured into a hipGraph; ``out`` reuses a previous plan's buffers."""
        rf = self.receptive_field
        grid = self.lift_grid(device)
        self.prebuilt_plan = ops.LiftPlan.build(grid, intrinsics[:, :rf], extrinsics[:, :rf], future_egomotion[:, :rf],
                                                self.encoder_out_channels, out=out)
        return self.prebuilt_plan

    def _bev_dtype(self):
        """float32 (stp3.py:230-232) -- or, under bf16 autocast with the channels-last BEV and a temporal model that runs
        its convolutions in bf16 anyway, bf16: the pooling kernel rounds its float32 sums once as it writes them (the
        cast the temporal model's first operator would make), half the bytes written, no cast pass."""
        if (self.bev_channels_last and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
                and isinstance(self.temporal_model, TemporalModel) and len(self.temporal_model.model) > 0):
            return torch.bfloat16
        return torch.float32

    def calculate_birds_eye_view_features(self, image, intrinsics, extrinsics, future_egomotion):
        """(B,S,N,3,H,W) images -> BEV features (B,S,C,X,Y) float32 + depth logits (B,S,N,D,fH,fW).
        Replaces stp3.py:303-318 (and everything it calls)."""
        b, s, n, c, h, w = image.shape
        dev = image.device
        grid = self.lift_grid(dev)
        if self.prebuilt_plan is not None:
            feat, depth = self.encoder(image.reshape(b * s * n, c, h, w))
            feat = feat.view(b, s, n, *feat.shape[1:])
            depth = depth.view(b, s, n, *depth.shape[1:])
            return ops.lift_splat(feat, depth, self.prebuilt_plan, self.discount, self.bev_channels_last,
                                  self._bev_dtype()), depth, self._cam_front(feat)
        # geometry-only work goes to a side stream: it overlaps the image encoder below
        cur = torch.cuda.current_stream(dev)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=dev)
        side = self._side_stream
        if intrinsics.is_cuda or extrinsics.is_cuda or future_egomotion.is_cuda:
            side.wait_stream(cur)           # pose tensors may still be in flight on the main stream
        with torch.cuda.stream(side):
            plan = ops.LiftPlan.build(grid, intrinsics, extrinsics, future_egomotion, self.encoder_out_channels)
        feat, depth = self.encoder(image.reshape(b * s * n, c, h, w))
        cur.wait_stream(side)
        # the plan's buffers were allocated on the side stream but are read by the pooling kernels (forward AND
        # backward) on the main stream: tell the allocator, or the next step's build may reuse them too early
        for buf in (plan.mats, plan.counts, plan.vox_cm, plan.plan):
            buf.record_stream(cur)
        feat = feat.view(b, s, n, *feat.shape[1:])
        depth = depth.view(b, s, n, *depth.shape[1:])
        bev = ops.lift_splat(feat, depth, plan, self.discount, self.bev_channels_last, self._bev_dtype())
        return bev, depth, self._cam_front(feat)

    def _cam_front(self, feat, cam_front_index=1):
        """The present frame's features of the front camera, (B, C, fH, fW): what the planner's GRU starts from
        (stp3.py:209-212 ``encoder_forward(cam_front_index=1)`` + :315 ``[:, -1]``); None without a planner."""
        return feat[:, -1, cam_front_index] if self.cfg.PLANNING.ENABLED else None

    def forward(self, image, intrinsics, extrinsics, future_egomotion):
        rf = self.receptive_field
        image = image[:, :rf].contiguous()
        intrinsics = intrinsics[:, :rf].contiguous()
        extrinsics = extrinsics[:, :rf].contiguous()
        future_egomotion = future_egomotion[:, :rf].contiguous()

        x, depth, cam_front = self.calculate_birds_eye_view_features(image, intrinsics, extrinsics, future_egomotion)
        output = {'depth_prediction': depth, 'cam_front': cam_front}

        if self.cfg.MODEL.TEMPORAL_MODEL.INPUT_EGOPOSE:
            # stp3.py:145-152: six broadcast ego-motion planes; frame 0 gets zeros, frame t gets ego[t-1].  The planes
            # are never built: every consumer of the temporal model's input is a 1x1x1 convolution or the whole-plane
            # pooling, so they enter the first block as per-frame constants (a bias of its fused BatchNorms)
            ego = future_egomotion.to(x.device, non_blocking=True).float()
            ego = torch.cat([torch.zeros_like(ego[:, :1]), ego[:, :rf - 1]], dim=1)
            if isinstance(self.temporal_model, TemporalModel) and len(self.temporal_model.model) > 0:
                states = self.temporal_model(x, ego)
            else:
                b, s, c = ego.shape
                states = self.temporal_model(torch.cat([x, ego.view(b, s, c, 1, 1).expand(b, s, c, *x.shape[-2:])], dim=2))
        else:
            states = self.temporal_model(x)
        if self.n_future > 0:
            # stp3.py:157-176: sample the present distribution, roll the present state forward, decode all frames
            present = states[:, -1:].contiguous()
            b, _, c, h, w = present.shape
            if self.cfg.PROBABILISTIC.ENABLED:
                dist = self.cfg.MODEL.DISTRIBUTION
                sample = self.distribution_forward(present, dist.MIN_LOG_SIGMA, dist.MAX_LOG_SIGMA)
            else:
                sample = present.new_zeros(b, 1, self.latent_dim, h, w)
            states = self.future_prediction(sample, states)
        output.update(self.decoder(states))
        return output

    def distribution_forward(self, present_features, min_log_sigma, max_log_sigma):
        """stp3.py:320-382: a sample of the present distribution, broadcast over the BEV plane
        (B,1,latent,H,W).  Training draws Gaussian noise, evaluation uses the mean."""
        b, s, _, h, w = present_features.shape
        assert s == 1
        latent, method = self.latent_dim, self.cfg.PROBABILISTIC.METHOD

        def sample_of(mu_log_sigma):
            mu = mu_log_sigma[:, :, :latent]
            sigma = torch.exp(torch.clamp(mu_log_sigma[:, :, latent:2 * latent], min_log_sigma, max_log_sigma))
            noise = (torch.randn((b, s, latent), device=present_features.device) if self.training
                     else torch.zeros((b, s, latent), device=present_features.device))
            return mu + sigma * noise.to(mu.dtype)

        if method == 'GAUSSIAN':
            sample = sample_of(self.present_distribution(present_features))
            return sample.view(b, s, latent, 1, 1).expand(b, s, latent, h, w)
        if method == 'BERNOULLI':
            log_prob = self.present_distribution(present_features)
            noise = (torch.randn((b, latent, h, w), device=present_features.device) if self.training
                     else torch.zeros((b, latent, h, w), device=present_features.device))
            return (torch.exp(log_prob) + noise.to(log_prob.dtype)).view(b, s, latent, h, w)
        if method == 'MIXGAUSSIAN':
            p = self.present_distribution(present_features)
            parts = [sample_of(p[:, :, 2 * i * latent:2 * (i + 1) * latent]) for i in range(3)]
            coef = torch.softmax(p[:, :, 6 * latent:], dim=-1)
            sample = parts[0] * coef[:, :, 0:1] + parts[1] * coef[:, :, 1:2] + parts[2] * coef[:, :, 2:3]
            return sample.view(b, s, latent, 1, 1).expand(b, s, latent, h, w)
        raise NotImplementedError(method)
