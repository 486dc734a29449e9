"""The planner's cost function -- drop-in for the reference's ``stp3/cost.py`` (same class names, parameters and call
signatures, so a reference checkpoint loads key for key: every term owns ``dx`` / ``bx`` parameters, the safety term its
``w``).

``Cost_Function.forward`` evaluates all seven terms for GPU tensors in ONE launch of ``stp3_traj_cost_fwd``
(csrc/stp3_plan.hip, through ``ops_plan.traj_cost``); the per-term modules below hold the torch statements of the same
arithmetic, which CPU tensors and float64 take (host-logic tests, the float64 truth) and which document each term.

``skimage.draw.polygon`` -- the reference's rasteriser of the ego box (cost.py:7,80; scikit-image is not installed in
this image) -- is restated in ``footprint_cells``: even-odd rule over the integer points of the bounding box, rows
first.  The published variants of that routine differ only for lattice points lying exactly on the outline, which
``footprint_cells`` rejects.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops_plan
from .utils import hp


def gen_dx_bx(xbound, ybound, zbound):
    """Resolution, first cell centre and cell count per axis as float32 / int64 tensors (stp3/utils/tools.py:176-181)."""
    rows = (xbound, ybound, zbound)
    return (torch.tensor([r[2] for r in rows], dtype=torch.float32),
            torch.tensor([r[0] + r[2] / 2.0 for r in rows], dtype=torch.float32),
            torch.tensor([int((r[1] - r[0]) / r[2]) for r in rows], dtype=torch.long))


def footprint_cells(corners_rc):
    """Integer (row, column) points inside the polygon with vertices ``corners_rc`` (float64 (V, 2)), in the order
    skimage.draw.polygon emits them (row by row, columns ascending).  Rule: the even-odd crossing test of
    scikit-image 0.18.1 -- the version the reference's environment.yml pins -- evaluated in float64
    (skimage/_shared/geometry.pxd ``point_in_polygon``: an edge is crossed by the rows r0 <= y < r1, half-open, and
    counts when the point lies strictly left of it).  A lattice point exactly ON the outline is therefore decided by
    that same half-open test (bottom / left edges in, top / right edges out for an axis-aligned box), as in the pinned
    release; releases from 0.19 on count every outline point as inside instead, which differs only for EGO / LIFT
    configurations whose box edges fall on cell boundaries (the default 1.85 m x 4.084 m box never does)."""
    r, c = np.asarray(corners_rc, dtype=np.float64).T
    rows = np.arange(int(max(0.0, r.min())), int(np.ceil(r.max())) + 1)
    cols = np.arange(int(max(0.0, c.min())), int(np.ceil(c.max())) + 1)
    yy, xx = np.meshgrid(rows.astype(np.float64), cols.astype(np.float64), indexing='ij')
    inside = np.zeros(yy.shape, dtype=bool)
    for i in range(len(r)):
        j = i - 1
        r0, c0, r1, c1 = r[i], c[i], r[j], c[j]
        if r0 == r1:
            continue                                      # a horizontal edge is never crossed by the half-open row test
        spans = ((r0 <= yy) & (yy < r1)) | ((r1 <= yy) & (yy < r0))
        with np.errstate(invalid='ignore', divide='ignore'):
            at = (c1 - c0) * (yy - r0) / (r1 - r0) + c0
        inside ^= spans & (xx < at)
    rr, cc = np.nonzero(inside)
    return np.stack([rows[rr], cols[cc]], axis=-1).astype(np.int64)


class BaseCost(nn.Module):
    """Grid constants and the footprint gathers shared by the terms (cost.py:51-163)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        dx, bx, nx = gen_dx_bx(cfg.LIFT.X_BOUND, cfg.LIFT.Y_BOUND, cfg.LIFT.Z_BOUND)
        self.dx = nn.Parameter(dx[:2], requires_grad=False)
        self.bx = nn.Parameter(bx[:2], requires_grad=False)
        self.bev_dimension = nx
        self.W = cfg.EGO.WIDTH
        self.H = cfg.EGO.HEIGHT
        self._cells = {}

    def footprint(self, lambda_=0):
        """(K, 2) int64 numpy (row, column) cells of the ego box inflated by ``lambda_`` metres; the box is centred
        0.5 m ahead of the ego origin (cost.py:70-83).  Depends on the configuration only: cached."""
        key = (lambda_, self.dx._version, self.bx._version)       # load_state_dict copies in place: the version moves
        if key not in self._cells:
            h, w = self.H, self.W
            pts = np.array([[-h / 2. + 0.5 - lambda_, w / 2. + lambda_], [h / 2. + 0.5 + lambda_, w / 2. + lambda_],
                            [h / 2. + 0.5 + lambda_, -w / 2. - lambda_], [-h / 2. + 0.5 - lambda_, -w / 2. - lambda_]])
            bx = self.bx.detach().cpu().numpy()
            dx = self.dx.detach().cpu().numpy()
            pts = (pts - bx) / dx                            # float64 - float32 -> float64, as in the reference
            self._cells = {k: v for k, v in self._cells.items() if k[1:] == key[1:]}      # drop stale grids
            self._cells[key] = footprint_cells(pts)          # (forward -> rows, lateral -> columns) already
        return self._cells[key]

    def get_origin_points(self, lambda_=0):
        return torch.from_numpy(self.footprint(lambda_)).to(device=self.bx.device)

    def get_points(self, trajs, lambda_=0):
        """Rows / columns (B, N, T, K) of the footprint cells of every trajectory point, clamped to the grid."""
        rc = self.get_origin_points(lambda_).to(trajs.device)
        scaled = trajs.unsqueeze(-2) / self.dx.to(trajs)
        rows = (scaled[..., 1] + rc[:, 0]).long().clamp(0, int(self.bev_dimension[0]) - 1)
        cols = (scaled[..., 0] + rc[:, 1]).long().clamp(0, int(self.bev_dimension[1]) - 1)
        return rows, cols

    def compute_area(self, semantic_pred, trajs, ego_velocity=None, _lambda=0):
        """Sum of ``semantic_pred`` (B, T, H, W) over the footprint at every trajectory point (x ``ego_velocity``)."""
        rows, cols = self.get_points(trajs, int(_lambda / self.dx[0]))
        B, N, T, _ = trajs.shape
        bi = torch.arange(B, device=trajs.device).view(B, 1, 1, 1)
        ti = torch.arange(T, device=trajs.device).view(1, 1, T, 1)
        area = semantic_pred[bi, ti, rows, cols].sum(dim=-1)
        return area if ego_velocity is None else area * ego_velocity

    def discretize(self, trajs):
        yi = ((trajs[..., 1] - self.bx[0].to(trajs)) / self.dx[0].to(trajs)).long().clamp(0, int(self.bev_dimension[0]) - 1)
        xi = ((trajs[..., 0] - self.bx[1].to(trajs)) / self.dx[1].to(trajs)).long().clamp(0, int(self.bev_dimension[1]) - 1)
        return yi, xi

    def evaluate(self, trajs, C):
        B, N, T, _ = trajs.shape
        yi, xi = self.discretize(trajs)
        return C[torch.arange(B, device=C.device).view(B, 1, 1), torch.arange(T, device=C.device).view(1, 1, T), yi, xi]


def _step_lengths(trajs):
    """|p_t - p_{t-1}| / 0.5 s with p_{-1} = origin: (B, N, T)."""
    prev = torch.cat([torch.zeros_like(trajs[:, :, :1]), trajs[:, :, :-1]], dim=2)
    return torch.sqrt(((trajs - prev) ** 2).sum(dim=-1)) / 0.5


def mask_from_map(m):
    """(B, 1, H, W) label map -> (B, H, W) as it is; (B, 2, H, W) logits -> softmax probability of class 1 (the
    caller zeroes what is below its threshold): the head of Rule / HeadwayCost / LR_divider.forward."""
    assert m.ndim == 4
    return torch.softmax(hp(m), dim=1)[:, 1] if m.shape[1] == 2 else m[:, 0]


def drivable_mask(drivable_area):
    """cost.py:196-201 / :258-263: probabilities below 0.5 are zeroed."""
    m = mask_from_map(drivable_area)
    return torch.where(m < 0.5, torch.zeros_like(m), m) if drivable_area.shape[1] == 2 else m


def lane_mask(lane_divider):
    """cost.py:289-294: probabilities up to and including 0.5 are zeroed."""
    m = mask_from_map(lane_divider)
    return torch.where(m <= 0.5, torch.zeros_like(m), m) if lane_divider.shape[1] == 2 else m


class Cost_Volume(BaseCost):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.factor = cfg.COST_FUNCTION.VOLUME

    def forward(self, trajs, cost_volume):
        return self.evaluate(trajs, torch.clamp(cost_volume, 0, 1000)) * self.factor


class Rule(BaseCost):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.factor = 5

    def forward(self, trajs, drivable_area):
        off_road = (drivable_mask(drivable_area) == 0).to(trajs.dtype)
        T = trajs.shape[2]
        return self.compute_area(off_road.unsqueeze(1).expand(-1, T, -1, -1), trajs) * self.factor


class SafetyCost(BaseCost):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.w = nn.Parameter(torch.tensor([1., 1.]), requires_grad=False)
        self._lambda = cfg.COST_FUNCTION.LAMBDA
        self.factor = cfg.COST_FUNCTION.SAFETY

    def forward(self, trajs, semantic_pred):
        speed = _step_lengths(trajs)
        under_box = self.compute_area(semantic_pred, trajs)
        near_box = self.compute_area(semantic_pred, trajs, speed, self._lambda)
        return (under_box * self.w[0].to(trajs) + near_box * self.w[1].to(trajs)) * self.factor


class HeadwayCost(BaseCost):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.L = 10
        self.factor = cfg.COST_FUNCTION.HEADWAY

    def forward(self, trajs, semantic_pred, drivable_area):
        on_road = semantic_pred * drivable_mask(drivable_area).unsqueeze(1)
        ahead = torch.stack([trajs[..., 0], trajs[..., 1] + self.L], dim=-1)
        return self.compute_area(on_road, ahead) * self.factor


class LR_divider(BaseCost):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.L = 1
        self.factor = cfg.COST_FUNCTION.LRDIVIDER

    def forward(self, trajs, lane_divider):
        lanes = lane_mask(lane_divider)
        yi, xi = self.discretize(trajs)
        cell = torch.stack([yi, xi], dim=-1)
        pitch = torch.flip(self.dx.to(trajs), dims=(0,))
        out = []
        for b in range(trajs.shape[0]):
            marks = torch.nonzero(lanes[b])
            if len(marks) == 0:
                out.append(torch.zeros(trajs.shape[1:3], device=trajs.device, dtype=trajs.dtype))
                continue
            dist = torch.sqrt((((cell[b].unsqueeze(-2) - marks) * pitch) ** 2).sum(dim=-1)).min(dim=-1).values
            out.append(torch.where(dist > self.L, torch.zeros_like(dist), (self.L - dist) ** 2))
        return torch.stack(out, dim=0) * self.factor


class Comfort(BaseCost):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.c_lat_acc = 3
        self.c_lon_acc = 3
        self.c_jerk = 1
        self.factor = cfg.COST_FUNCTION.COMFORT

    def forward(self, trajs):
        prev = torch.cat([torch.zeros_like(trajs[:, :, :1]), trajs[:, :, :-1]], dim=2)
        vel = (trajs - prev) / 0.5                                         # (lateral, longitudinal) per step
        acc = torch.zeros_like(vel)
        acc[:, :, 1:] = (vel[:, :, 1:] - vel[:, :, :-1]) / 0.5
        peak = acc.abs().max(dim=2).values                                 # (B, N, 2)
        speed = _step_lengths(trajs)
        sacc = torch.zeros_like(speed)
        sacc[:, :, 1:] = (speed[:, :, 1:] - speed[:, :, :-1]) / 0.5
        jerk = torch.zeros_like(speed)
        jerk[:, :, 2:] = (sacc[:, :, 2:] - sacc[:, :, 1:-1]) / 0.5
        jerk = jerk.abs().max(dim=-1).values
        cost = torch.clamp(peak[..., 0] - self.c_lat_acc, 0, 30) ** 2
        cost = cost + torch.clamp(peak[..., 1] - self.c_lon_acc, 0, 30) ** 2
        cost = cost + torch.clamp(jerk - self.c_jerk, 0, 20) ** 2
        return cost * self.factor


class Progress(BaseCost):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.factor = cfg.COST_FUNCTION.PROGRESS

    def forward(self, trajs, target_points):
        farthest = trajs[..., 1].max(dim=-1).values
        to_goal = ((trajs[:, :, -1] - target_points.to(trajs).unsqueeze(1)) ** 2).sum(dim=-1)
        # the reference branches on the batch's summed target (a host synchronisation there); a select here
        use_goal = (target_points.sum() >= 0.5).to(trajs.dtype)
        return (to_goal * use_goal - farthest) * self.factor


class Cost_Function(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.safetycost = SafetyCost(cfg)
        self.headwaycost = HeadwayCost(cfg)
        self.lrdividercost = LR_divider(cfg)
        self.comfortcost = Comfort(cfg)
        self.progresscost = Progress(cfg)
        self.rulecost = Rule(cfg)
        self.costvolume = Cost_Volume(cfg)
        self.n_future = cfg.N_FUTURE_FRAMES
        self._tables = {}

    def _kernel_inputs(self, device):
        """Footprint tables on the device and the scalar fields of ``stp3_plan_dims``, cached per device and per VALUE
        of the parameters they are cut from (their version counters: a ``load_state_dict`` after the first forward must
        not leave the kernel path on the old grid / weights while the torch statements read the new ones)."""
        s = self.safetycost
        key = (str(device), s.dx._version, s.bx._version, s.w._version)
        if key not in self._tables:
            self._tables = {k: v for k, v in self._tables.items() if k[1:] == key[1:]}
            inflate = int(s._lambda / float(s.dx[0]))
            fp0 = torch.from_numpy(s.footprint(0)).to(device=device, dtype=torch.int32).contiguous()
            fpl = torch.from_numpy(s.footprint(inflate)).to(device=device, dtype=torch.int32).contiguous()
            dx, bx, w = s.dx.detach().cpu(), s.bx.detach().cpu(), s.w.detach().cpu()
            dx, bx, w = [float(v) for v in dx], [float(v) for v in bx], [float(v) for v in w]      # values, not views
            params = dict(dx0=dx[0], dx1=dx[1], bx0=bx[0], bx1=bx[1], safety=s.factor, headway=self.headwaycost.factor,
                          lrdivider=self.lrdividercost.factor, comfort=self.comfortcost.factor,
                          progress=self.progresscost.factor, volume=self.costvolume.factor, rule=self.rulecost.factor,
                          w0=w[0], w1=w[1], headway_dist=self.headwaycost.L, lr_dist=self.lrdividercost.L)
            self._tables[key] = (fp0, fpl, params)
        return self._tables[key]

    def forward(self, cost_volume, trajs, semantic_pred, lane_divider, drivable_area, target_point):
        """cost_volume (B, T, H, W); trajs (B, N, T, 2) metres; semantic_pred (B, T, H, W); lane_divider /
        drivable_area (B, 1 | 2, H, W); target_point (B, 2) -> cost_fc (B, N), cost_fo (B, N, T)   (cost.py:26-47)"""
        if ops_plan.supported(cost_volume, trajs):
            assert tuple(cost_volume.shape[-2:]) == tuple(int(v) for v in self.safetycost.bev_dimension[:2])
            fp0, fpl, params = self._kernel_inputs(cost_volume.device)
            return ops_plan.traj_cost(cost_volume, trajs, semantic_pred, drivable_mask(drivable_area),
                                      lane_mask(lane_divider), target_point, fp0, fpl, params)
        trajs = trajs * torch.tensor([-1, 1], device=trajs.device, dtype=trajs.dtype)
        sem = semantic_pred.to(trajs.dtype)
        safety = torch.clamp(self.safetycost(trajs, sem), 0, 100)
        headway = torch.clamp(self.headwaycost(trajs, sem, drivable_area), 0, 100)
        lrdivider = torch.clamp(self.lrdividercost(trajs, lane_divider), 0, 100)
        comfort = torch.clamp(self.comfortcost(trajs), 0, 100)
        progress = torch.clamp(self.progresscost(trajs, target_point), -100, 100)
        rule = torch.clamp(self.rulecost(trajs, drivable_area), 0, 100)
        volume = torch.clamp(self.costvolume(trajs, cost_volume), 0, 100)
        return comfort + progress, safety + headway + lrdivider + volume + rule
