"""Host-side operators of the lift / voxel-pool path: thin wrappers that hand raw device pointers
to libstp3hip.so (include/stp3_hip.h) on the caller's current HIP stream.

Mirrors the reference's operator surface for this path:
  * ``lift_matrices``      -- the tiny host-side part of ``STP3.get_geometry`` (stp3/models/stp3.py:189,196)
                              and ``pose_vec2mat`` (stp3/utils/geometry.py:124-172)
  * ``LiftPlan.build``     -- ``get_geometry`` + ego alignment + index + mask + sort
                              (stp3.py:192-198, 270-277, 287-289, 239-257), geometry only
  * ``lift_splat``         -- softmax(depth) (x) feat, ``VoxelsSumming`` and the discounted
                              accumulation (stp3.py:215-221, geometry.py:299-330, stp3.py:279-299),
                              differentiable (``torch.autograd.Function``)
PyTorch is used for device memory, streams and autograd plumbing only.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import LiftDims, VOX_PIXELMAJOR, VOX_REFERENCE, check


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws_key(device):
    """Key of a scratch buffer: one per device AND stream -- operators of independent branches are queued on different HIP
    streams (the encoder's two heads, the weight gradients beside the data gradients) and must not share scratch."""
    device = torch.device(device)
    if device.type != 'cuda':
        return device
    return (device, torch.cuda.current_stream().cuda_stream)


def _stream_handle():
    return torch.cuda.current_stream().cuda_stream


# Optional live timing of the HIP operators with events recorded on the stream they are launched on
# (bench.py's roofline numbers).  Off by default: zero overhead in normal use.
PROFILE_ENABLED = False
PROFILE = []          # (name, start_event, end_event)


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PROFILE_ENABLED:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if PROFILE_ENABLED:
            self.end.record()
            PROFILE.append((self.name, self.start, self.end))
        return False


def profile_summary():
    """name -> {'n', 'avg_ms'} over everything recorded since PROFILE was cleared (synchronises)."""
    torch.cuda.synchronize()
    acc = {}
    for name, s, e in PROFILE:
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += s.elapsed_time(e)
    return {k: {'n': n, 'avg_ms': t / n} for k, (n, t) in acc.items()}


def _fast_apply(cls):
    """``cls.apply`` minus the Python prologue of ``torch.autograd.Function.apply`` (setup_context probing and the
    functorch dead-wrapper scan over every argument): what that method itself calls when no functorch transform is
    active.  ~6 us per call on the host, ~300 custom-operator calls per training step."""
    return super(torch.autograd.Function, cls).apply


def _need_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise _lib.Stp3HipError('stp3_amd operators run on the GPU only (got a CPU tensor); there is no fallback')


def make_dims(B, T, N, D, fH, fW, C, X, Y, Z):
    return LiftDims(int(B), int(T), int(N), int(D), int(fH), int(fW), int(C), int(X), int(Y), int(Z))


# ----------------------------------------------------------------------------------------------
# host-side constants
# ----------------------------------------------------------------------------------------------
def lift_matrices(intrinsics, extrinsics, future_egomotion):
    """Camera and ego matrices, built on the HOST with the same torch CPU ops the reference uses,
    so that libm / LAPACK differences of a device implementation can never move a point across a
    voxel border (hard part 2).

    intrinsics (B,S,N,3,3), extrinsics (B,S,N,4,4), future_egomotion (B,S,6): CPU or GPU tensors
    (GPU tensors are copied back; these are a few hundred floats).
    Returns float32 CPU tensors cam_m (B*S*N,9), cam_t (B*S*N,3), ego_r (B*S,9), ego_t (B*S,3).
    """
    intr = intrinsics.detach().float().cpu()
    extr = extrinsics.detach().float().cpu()
    ego = future_egomotion.detach().float().cpu()
    # stp3.py:189,196  combined_transformation = rotation.matmul(torch.inverse(intrinsics))
    cam_m = extr[..., :3, :3].matmul(torch.inverse(intr)).reshape(-1, 9).contiguous()
    cam_t = extr[..., :3, 3].reshape(-1, 3).contiguous()
    # geometry.py:124-155  R = X(rx) . Y(ry) . Z(rz)
    ang = ego[..., 3:].reshape(-1, 3)
    rx, ry, rz = ang[:, 0], ang[:, 1], ang[:, 2]
    zero, one = torch.zeros_like(rz), torch.ones_like(rz)
    cz, sz = torch.cos(rz), torch.sin(rz)
    zmat = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).view(-1, 3, 3)
    cy, sy = torch.cos(ry), torch.sin(ry)
    ymat = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).view(-1, 3, 3)
    cx, sx = torch.cos(rx), torch.sin(rx)
    xmat = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).view(-1, 3, 3)
    ego_r = xmat.bmm(ymat).bmm(zmat).reshape(-1, 9).contiguous()
    ego_t = ego[..., :3].reshape(-1, 3).contiguous()
    return cam_m, cam_t, ego_r, ego_t


def separable_frustum(frustum):
    """(D,fH,fW,3) frustum parameter (stp3.py:111-130) -> xs (fW), ys (fH), ds (D).
    ``create_frustum`` only ever builds separable grids; anything else is rejected."""
    fr = frustum.detach().float().cpu()
    xs, ys, ds = fr[0, 0, :, 0].clone(), fr[0, :, 0, 1].clone(), fr[:, 0, 0, 2].clone()
    d_, fh, fw, _ = fr.shape
    ok = (torch.equal(fr[..., 0], xs.view(1, 1, fw).expand(d_, fh, fw))
          and torch.equal(fr[..., 1], ys.view(1, fh, 1).expand(d_, fh, fw))
          and torch.equal(fr[..., 2], ds.view(d_, 1, 1).expand(d_, fh, fw)))
    if not ok:
        raise _lib.Stp3HipError('frustum is not separable into (x[w], y[h], depth[d]); unsupported')
    return xs, ys, ds


class LiftGrid:
    """Device-resident constants of one model: frustum axes and BEV grid (uploaded once)."""

    def __init__(self, frustum, bev_resolution, bev_start_position, bev_dimension, device):
        xs, ys, ds = separable_frustum(frustum)
        self.D, self.fH, self.fW = ds.numel(), ys.numel(), xs.numel()
        res = bev_resolution.detach().float().cpu()
        start = bev_start_position.detach().float().cpu()
        # stp3.py:288  (bev_start_position - bev_resolution / 2.0), in float32
        off = start - res / 2.0
        self.X, self.Y, self.Z = (int(v) for v in bev_dimension.detach().cpu().tolist())
        self.device = torch.device(device)
        self.consts = torch.cat([xs, ys, ds, off, res]).to(self.device)
        o = 0
        self.xs = self.consts[o:o + self.fW]; o += self.fW
        self.ys = self.consts[o:o + self.fH]; o += self.fH
        self.ds = self.consts[o:o + self.D]; o += self.D
        self.off = self.consts[o:o + 3]; o += 3
        self.res = self.consts[o:o + 3]


def voxel_index(grid, dims, cam_m, cam_t, ego_r, ego_t, order=VOX_REFERENCE, counts=None, out=None):
    """Raw ``stp3_voxel_index``: int32 voxel ids [B*T, P] in ``order``; cam/ego matrices are
    device float32 tensors laid out as ``lift_matrices`` returns them."""
    _need_gpu(cam_m, cam_t, ego_r, ego_t)
    vox = out if out is not None else torch.empty(dims.BT, dims.P, dtype=torch.int32, device=cam_m.device)
    rc = _lib.lib().stp3_voxel_index(ctypes.byref(dims), _ptr(cam_m), _ptr(cam_t), _ptr(ego_r), _ptr(ego_t),
                                     _ptr(grid.xs), _ptr(grid.ys), _ptr(grid.ds), _ptr(grid.off), _ptr(grid.res),
                                     int(order), _ptr(vox), _ptr(counts) if counts is not None else None, _stream())
    check(rc, 'stp3_voxel_index')
    return vox


class PinnedUpload:
    """Asynchronous host -> device refresh of a small static device tensor (a hipGraph's inputs that depend on the batch's
    poses: camera / ego matrices, label-warp matrices).  A copy from pageable memory is staged synchronously -- the host
    would wait for everything queued before it -- so the values go through one of TWO pinned buffers; a buffer is reused
    only after the copy that last read it has completed (its event), which bounds the host's lead to two refreshes."""

    def __init__(self, dst):
        self.dst = dst
        self.pinned = dst.is_cuda and torch.cuda.is_available()      # (the CPU dry runs pose as a GPU without one)
        if self.pinned:
            self.bufs = [torch.empty(dst.shape, dtype=dst.dtype, pin_memory=True) for _ in range(2)]
            self.events = [None, None]
            self.k = 0

    def __call__(self, src):
        if not self.pinned:
            self.dst.copy_(src)
            return
        k, self.k = self.k, self.k ^ 1
        if self.events[k] is not None:
            self.events[k].synchronize()
        self.bufs[k].copy_(src.reshape(self.dst.shape))
        self.dst.copy_(self.bufs[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dst.device))
        self.events[k] = ev


class LiftPlan:
    """Geometry-only pooling plan for one batch: voxel ids (``vox_cm``: column-major [B*T, N*fW, D, fH]) + per-voxel
    lists of column runs.

    Built from camera/ego poses alone, i.e. independent of the image encoder -- ``build`` can be
    issued on a side stream while the encoder runs (see ``STP3.forward``).
    """

    def __init__(self, dims, vox_cm, plan, counts):
        self.dims, self.vox_cm, self.plan, self.counts = dims, vox_cm, plan, counts

    @staticmethod
    def build(grid, intrinsics, extrinsics, future_egomotion, channels, out=None):
        """``out``: a plan of the same shape whose device buffers are overwritten in place (static addresses)."""
        b, s, n = intrinsics.shape[:3]
        dims = make_dims(b, s, n, grid.D, grid.fH, grid.fW, channels, grid.X, grid.Y, grid.Z)
        mats = torch.cat([m.reshape(-1) for m in lift_matrices(intrinsics, extrinsics, future_egomotion)])
        lib = _lib.lib()
        if out is not None:
            assert bytes(out.dims) == bytes(dims), 'LiftPlan.build(out=...): shape changed'
            if getattr(out, 'mats_upload', None) is None or out.mats_upload.dst is not out.mats:
                out.mats_upload = PinnedUpload(out.mats)
            out.mats_upload(mats)                   # (asynchronous: pinned double buffer)
            mats, counts, vox_cm, plan, nbytes = out.mats, out.counts, out.vox_cm, out.plan, out.plan.numel()
        else:
            mats = mats.to(grid.device, non_blocking=True)
            # run counters: zero once; every build counts them up and back down to zero
            counts = torch.zeros(dims.BT, dims.V, dtype=torch.int32, device=grid.device)
            vox_cm = torch.empty(dims.BT, dims.P, dtype=torch.int32, device=grid.device)
            size = ctypes.c_size_t()
            check(lib.stp3_lift_plan_bytes(ctypes.byref(dims), ctypes.byref(size)), 'stp3_lift_plan_bytes')
            nbytes = size.value
            plan = torch.empty(nbytes, dtype=torch.uint8, device=grid.device)
        _need_gpu(mats)
        n_cam = b * s * n
        cam_m = mats[:n_cam * 9]
        cam_t = mats[n_cam * 9:n_cam * 12]
        ego_r = mats[n_cam * 12:n_cam * 12 + b * s * 9]
        ego_t = mats[n_cam * 12 + b * s * 9:]
        with _timed('plan_build'):
            rc = lib.stp3_lift_plan_build(ctypes.byref(dims), _ptr(cam_m), _ptr(cam_t), _ptr(ego_r), _ptr(ego_t),
                                          _ptr(grid.xs), _ptr(grid.ys), _ptr(grid.ds), _ptr(grid.off), _ptr(grid.res),
                                          _ptr(vox_cm), _ptr(counts), _ptr(plan), ctypes.c_size_t(nbytes), _stream())
        check(rc, 'stp3_lift_plan_build')
        res = out if out is not None else LiftPlan(dims, vox_cm, plan, counts)
        res.mats = mats
        return res

    @staticmethod
    def _align256(n):
        return (n + 255) & ~255

    def voxel_ids(self):
        """The ids as a (B, T, N, D, fH, fW) view -- the reference's index order (stp3.py:284)."""
        d = self.dims
        return self.vox_cm.view(d.B, d.T, d.N, d.fW, d.D, d.fH).permute(0, 1, 2, 4, 5, 3)

    def _sections(self):
        """Byte offsets of the plan's sections (include/stp3_hip.h: vox_off, masks, col_cnt, col_off, tmp, vox_runs,
        run_desc, run_vox)."""
        d = self.dims
        a = self._align256
        ncol = d.BT * d.N * d.fW
        sizes = [a(d.BT * (d.V + 1) * 4), a(ncol * d.fH * 16), a(ncol * 4), a((ncol + 1) * 4), a(d.BT * d.P * 4),
                 a(d.BT * d.P * 4), a(d.BT * d.P * 4), a(d.BT * d.P * 4)]
        return [sum(sizes[:i]) for i in range(len(sizes))]

    def offsets(self):
        """[BT, V+1] int32 view: exclusive scan of the runs per voxel (per frame)."""
        d = self.dims
        return self.plan[:d.BT * (d.V + 1) * 4].view(torch.int32).view(d.BT, d.V + 1)

    def masks(self):
        """[BT, N*fW, fH, 2] int64 view: bit d of [..., 0] = a run of depth bin d ends at this row of the image column,
        bit d of [..., 1] = point (d, h) lies inside the BEV grid."""
        d = self.dims
        o = self._sections()[1]
        return self.plan[o:o + d.BT * d.N * d.fW * d.fH * 16].view(torch.int64).view(d.BT, d.N * d.fW, d.fH, 2)

    def column_offsets(self):
        """[BT*N*fW + 1] int32 view: slot of every image column's first run (exclusive scan over all frames); the
        last entry is the total number of runs."""
        d = self.dims
        o = self._sections()[3]
        return self.plan[o:o + (d.BT * d.N * d.fW + 1) * 4].view(torch.int32)

    def run_descriptors(self):
        """[BT*P] int32 view, one word per slot: depth bin | first row << 8 | last row << 16 of the run."""
        d = self.dims
        o = self._sections()[6]
        return self.plan[o:o + d.BT * d.P * 4].view(torch.int32)

    def run_voxels(self):
        """[BT*P] int32 view, one word per slot: the voxel the run falls into."""
        d = self.dims
        o = self._sections()[7]
        return self.plan[o:o + d.BT * d.P * 4].view(torch.int32)

    def run_lists(self):
        """[BT*P] int32 view: the per-voxel slot lists; voxel v of frame bt owns
        ``[column_offsets()[bt*N*fW] + offsets()[bt, v], ... + offsets()[bt, v+1])``, ascending."""
        d = self.dims
        o = self._sections()[5]
        return self.plan[o:o + d.BT * d.P * 4].view(torch.int32)


def depth_softmax(dims, logits_pm):
    """logits [BT, NPIX, D] (pixel-major) -> probabilities [BT, N*fW, D, fH] (column-major, see include/stp3_hip.h)."""
    _need_gpu(logits_pm)
    prob = torch.empty(dims.BT, dims.N * dims.fW, dims.D, dims.fH, dtype=torch.float32, device=logits_pm.device)
    check(_lib.lib().stp3_depth_softmax(ctypes.byref(dims), _ptr(logits_pm), _ptr(prob), _stream()),
          'stp3_depth_softmax')
    return prob


BEV_CHANNELS_FIRST, BEV_CHANNELS_LAST = _lib.BEV_CHANNELS_FIRST, _lib.BEV_CHANNELS_LAST
_WORKSPACE = {}


def lift_workspace(dims, device):
    """Scratch of the pooling calls (one buffer per device, grown on demand): the run slots of the forward, the
    voxel-major gradient G_t of the backward and, for the reference (channels-first) layout, the channels-last BEV
    before the transpose pass."""
    nbytes = ctypes.c_size_t()
    check(_lib.lib().stp3_lift_workspace_bytes(ctypes.byref(dims), ctypes.byref(nbytes)), 'stp3_lift_workspace_bytes')
    key = _ws_key(device)
    ws = _WORKSPACE.get(key)
    if ws is None or ws.numel() < nbytes.value:
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=torch.device(device))
        _WORKSPACE[key] = ws
    return ws, nbytes.value


class _LiftSplat(torch.autograd.Function):
    """feat_pm [BT,NPIX,C], logits_pm [BT,NPIX,D] (float32, contiguous) -> bev of logical shape [B,T,C,X,Y];
    ``channels_last``: its memory is [B,T,X,Y,C] (no transpose passes), else the reference's [B,T,C,X,Y]."""

    @staticmethod
    def forward(ctx, feat_pm, logits_pm, lift_plan, discount, channels_last, bf16_out=False):
        _need_gpu(feat_pm, logits_pm)
        d = lift_plan.dims
        feat_pm = feat_pm.contiguous()
        logits_pm = logits_pm.contiguous()
        assert feat_pm.dtype == torch.float32 and logits_pm.dtype == torch.float32
        assert feat_pm.shape == (d.BT, d.NPIX, d.C) and logits_pm.shape == (d.BT, d.NPIX, d.D)
        dev = feat_pm.device
        # the probabilities are a by-product of the forward kernel (column-major); only the backward kernel of the
        # general shapes reads them -- the matrix-core one recomputes them from the logits
        needs = ctypes.c_int()
        check(_lib.lib().stp3_lift_bwd_needs_prob(ctypes.byref(d), ctypes.byref(needs)), 'stp3_lift_bwd_needs_prob')
        prob = (torch.empty(d.BT, d.N * d.fW, d.D, d.fH, dtype=torch.float32, device=dev)
                if needs.value and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None)
        if bf16_out and not channels_last:
            raise _lib.Stp3HipError('lift_splat: the bf16 BEV output exists for the channels-last layout only')
        layout = (_lib.BEV_CHANNELS_LAST_BF16 if bf16_out else BEV_CHANNELS_LAST) if channels_last else BEV_CHANNELS_FIRST
        shape = (d.B, d.T, d.X, d.Y, d.C) if channels_last else (d.B, d.T, d.C, d.X, d.Y)
        bev = torch.empty(shape, dtype=torch.bfloat16 if bf16_out else torch.float32, device=dev)
        ws, ws_bytes = lift_workspace(d, dev)
        ws_ptr = _ptr(ws)
        with _timed('lift_splat_fwd'):
            rc = _lib.lib().stp3_lift_splat_fwd(ctypes.byref(d), _ptr(feat_pm), _ptr(logits_pm), _ptr(lift_plan.plan),
                                                ctypes.c_float(discount), layout, ws_ptr, ctypes.c_size_t(ws_bytes),
                                                _ptr(prob) if prob is not None else None, _ptr(bev), _stream())
        check(rc, 'stp3_lift_splat_fwd')
        ctx.save_for_backward(feat_pm, logits_pm, prob)
        ctx.lift_plan = lift_plan
        ctx.discount = discount
        ctx.channels_last = channels_last
        return bev.permute(0, 1, 4, 2, 3) if channels_last else bev

    @staticmethod
    def backward(ctx, grad_bev):
        feat_pm, logits_pm, prob = ctx.saved_tensors
        d = ctx.lift_plan.dims
        if ctx.channels_last:
            # taken as it comes when it already is [B,T,X,Y,C] memory in float32 / bfloat16; copied once otherwise
            if grad_bev.dtype not in (torch.float32, torch.bfloat16):
                grad_bev = grad_bev.float()
            grad_bev = grad_bev.permute(0, 1, 3, 4, 2).contiguous()
            layout = BEV_CHANNELS_LAST
        else:
            grad_bev = grad_bev.contiguous().float()
            layout = BEV_CHANNELS_FIRST
        gdt = _lib.DTYPE_BF16 if grad_bev.dtype == torch.bfloat16 else _lib.DTYPE_F32
        ws, ws_bytes = lift_workspace(d, grad_bev.device)
        grad_feat = torch.empty_like(feat_pm)
        grad_logits = torch.empty(d.BT, d.NPIX, d.D, dtype=torch.float32, device=feat_pm.device)
        with _timed('lift_splat_bwd'):
            rc = _lib.lib().stp3_lift_splat_bwd(ctypes.byref(d), _ptr(grad_bev), layout, gdt, _ptr(feat_pm),
                                                _ptr(logits_pm), _ptr(prob) if prob is not None else None,
                                                _ptr(ctx.lift_plan.vox_cm), _ptr(ctx.lift_plan.plan),
                                                ctypes.c_float(ctx.discount), _ptr(ws), ctypes.c_size_t(ws_bytes),
                                                _ptr(grad_feat), _ptr(grad_logits), _stream())
        check(rc, 'stp3_lift_splat_bwd')
        return grad_feat, grad_logits, None, None, None, None


def lift_splat(feat, depth_logits, lift_plan, discount, channels_last=False, out_dtype=torch.float32):
    """Differentiable lift + voxel pool.

    feat (B,T,N,C,fH,fW) and depth_logits (B,T,N,D,fH,fW) in any memory format (channels-last
    memory makes the re-layout free); returns the BEV features (B,T,C,X,Y) float32 (always float32, even
    under autocast: stp3.py:230-232).  ``channels_last=False``: contiguous, the reference's layout;
    ``True``: the same logical tensor stored [B,T,X,Y,C] -- what the NHWC convolutions downstream read and
    what their gradient arrives in -- with no transpose pass on either side.  ``out_dtype=torch.bfloat16`` (channels-last
    only): the float32 sums are rounded once to bf16 as they are written -- the values a bf16 consumer would make of the
    float32 tensor, without the tensor and without the consumer's cast pass; everything up to that rounding (softmax,
    run sums, per-voxel sums, the discounted accumulation over the frames) stays float32."""
    d = lift_plan.dims
    feat_pm = feat.float().permute(0, 1, 2, 4, 5, 3).reshape(d.BT, d.NPIX, d.C)
    logits_pm = depth_logits.float().permute(0, 1, 2, 4, 5, 3).reshape(d.BT, d.NPIX, d.D)
    return _LiftSplat.apply(feat_pm, logits_pm, lift_plan, float(discount), bool(channels_last),
                            out_dtype == torch.bfloat16)


# ----------------------------------------------------------------------------------------------
# stand-alone voxel summing (compatibility with the reference's operator boundary)
# ----------------------------------------------------------------------------------------------
class VoxelsSumming(torch.autograd.Function):
    """Drop-in for the reference's ``VoxelsSumming`` (stp3/utils/geometry.py:299-330).

    ``VoxelsSumming.apply(x (M,C) float32 sorted by rank, geometry (M,3), ranks (M,)) ->
    (x_sum (V',C), geometry_kept (V',3))``: one output row per distinct consecutive rank, the
    geometry of the LAST point of each voxel (the rows the reference's boolean mask keeps,
    geometry.py:308-311).  The model itself uses the fused ``lift_splat``; this operator is for
    callers that hold the sorted point matrix.  Like the reference's boolean indexing, deriving
    the number of voxels synchronises with the host once."""

    @staticmethod
    def forward(ctx, x, geometry, ranks):
        _need_gpu(x, geometry, ranks)
        if x.dim() != 2 or ranks.dim() != 1 or ranks.shape[0] != x.shape[0] or geometry.shape[0] != x.shape[0]:
            raise _lib.Stp3HipError('VoxelsSumming expects x (M,C), geometry (M,...), ranks (M,)')
        m, c = x.shape
        xf = x.contiguous().float()
        kept = torch.ones(m, dtype=torch.bool, device=x.device)
        if m > 1:
            kept[:-1] = ranks[1:] != ranks[:-1]
        ends = kept.nonzero().flatten()                         # last row of every voxel (host sync)
        seg_off = torch.zeros(ends.numel() + 1, dtype=torch.int32, device=x.device)
        seg_off[1:] = ends + 1
        out = torch.empty(ends.numel(), c, dtype=torch.float32, device=x.device)
        check(_lib.lib().stp3_voxels_sum_fwd(_ptr(xf), _ptr(seg_off), int(ends.numel()), int(c), _ptr(out), _stream()),
              'stp3_voxels_sum_fwd')
        geometry_kept = geometry[kept]
        ctx.save_for_backward(seg_off)
        ctx.rows, ctx.x_dtype = m, x.dtype
        ctx.mark_non_differentiable(geometry_kept)
        return out.to(x.dtype), geometry_kept

    @staticmethod
    def backward(ctx, grad_x, grad_geometry):
        (seg_off,) = ctx.saved_tensors
        g = grad_x.contiguous().float()
        out = torch.empty(ctx.rows, g.shape[1], dtype=torch.float32, device=g.device)
        check(_lib.lib().stp3_voxels_sum_bwd(_ptr(g), _ptr(seg_off), int(seg_off.numel() - 1), int(g.shape[1]),
                                             _ptr(out), _stream()), 'stp3_voxels_sum_bwd')
        return out.to(ctx.x_dtype), None, None


# ----------------------------------------------------------------------------------------------
# depthwise convolution (EfficientNet MBConv blocks)
# ----------------------------------------------------------------------------------------------
def _dw_dims(x, k, stride, pad_top, pad_left, ho, wo):
    n, c, h, w = x.shape
    if x.dtype == torch.bfloat16:
        dt = _lib.DTYPE_BF16
    elif x.dtype == torch.float32:
        dt = _lib.DTYPE_F32
    else:
        raise _lib.Stp3HipError(f'depthwise conv supports float32 / bfloat16, got {x.dtype}')
    return _lib.DwConvDims(n, h, w, c, ho, wo, k, stride, pad_top, pad_left, dt)


_DW_WEIGHT_CACHE = {}


def _dw_weight_taps(weight):
    """(C,1,K,K) depthwise weight -> [K*K][C] float32 (tap-major: what the kernels read, 16-byte vectors along C).  The
    transposed copy of a PARAMETER is kept until the parameter changes (its version counter, or the epoch of
    out-of-band updates ``invalidate_weight_cache`` bumps): one tiny transpose kernel per layer and optimizer step
    instead of one per forward AND backward use."""
    c, _, k, _ = weight.shape
    if not (isinstance(weight, torch.nn.Parameter) and weight.is_leaf):
        return weight.detach().float().reshape(c, k * k).t().contiguous()
    if ASSEMBLED_WEIGHTS and weight.is_cuda and weight.dtype == torch.float32:
        # with the convolution weights' shadows: rewritten by the one launch of stp3_conv2d_prep_weights after an optimizer step
        tab = _shadows(weight.device)
        ent = tab.lookup(weight) or tab.register_taps(weight)
        return ent['taps']
    key = id(weight)
    stamp = (weight._version, _WEIGHT_EPOCH[0], weight.data_ptr(), tuple(weight.shape))
    ent = _DW_WEIGHT_CACHE.get(key)
    if ent is None or ent[0]() is not weight or ent[1] != stamp:
        wt = weight.detach().float().reshape(c, k * k).t().contiguous()
        if ent is None or ent[0]() is not weight:
            # one finalizer per parameter OBJECT, not one per optimizer step (they would pile up in
            # weakref.finalize's registry for the whole run)
            weakref.finalize(weight, _DW_WEIGHT_CACHE.pop, key, None)
        ent = (weakref.ref(weight), stamp, wt)
        _DW_WEIGHT_CACHE[key] = ent
    return ent[2]


class _DepthwiseConv2d(torch.autograd.Function):
    """x (N,C,H,W) channels-last memory, weight (C,1,K,K) [, bias (C)] -> y (N,C,Ho,Wo) channels-last memory.
    pad = (left, right, top, bottom) explicit zero padding ("static same": may be asymmetric)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, bias=None):
        _need_gpu(x, weight)
        c, _, k, _ = weight.shape
        left, right, top, bottom = pad
        n, _, h, w = x.shape
        ho = (h + top + bottom - k) // stride + 1
        wo = (w + left + right - k) // stride + 1
        x = x.contiguous(memory_format=torch.channels_last)
        wt = _dw_weight_taps(weight)                                            # [K*K][C] float32
        y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dims = _dw_dims(x, k, stride, top, left, ho, wo)
        if bias is None:
            check(_lib.lib().stp3_dwconv2d_fwd(ctypes.byref(dims), _ptr(x), _ptr(wt), _ptr(y), _stream()),
                  'stp3_dwconv2d_fwd')
        else:
            bf = bias.detach().float().contiguous()
            check(_lib.lib().stp3_dwconv2d_fwd_bias(ctypes.byref(dims), _ptr(x), _ptr(wt), _ptr(bf), _ptr(y), _stream()),
                  'stp3_dwconv2d_fwd_bias')
        ctx.save_for_backward(x, wt)
        ctx.dims = dims
        ctx.wshape = weight.shape
        ctx.wdtype = weight.dtype
        ctx.bdtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        dims = ctx.dims
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = dw = None
        lib = _lib.lib()
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            check(lib.stp3_dwconv2d_bwd_data(ctypes.byref(dims), _ptr(dy), _ptr(wt), _ptr(dx), _stream()),
                  'stp3_dwconv2d_bwd_data')
        if ctx.needs_input_grad[1]:
            nbytes = ctypes.c_size_t()
            check(lib.stp3_dwconv2d_bwd_weight_workspace(ctypes.byref(dims), ctypes.byref(nbytes)),
                  'stp3_dwconv2d_bwd_weight_workspace')
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=x.device)
            dwt = torch.empty_like(wt)
            check(lib.stp3_dwconv2d_bwd_weight(ctypes.byref(dims), _ptr(x), _ptr(dy), _ptr(dwt), _ptr(ws),
                                               ctypes.c_size_t(nbytes.value), _stream()), 'stp3_dwconv2d_bwd_weight')
            dw = dwt.t().reshape(ctx.wshape).to(ctx.wdtype)
        db = None
        if ctx.bdtype is not None and ctx.needs_input_grad[4]:
            db = channel_sums(dy).to(ctx.bdtype)
        return dx, dw, None, None, db


_DW_APPLY = _fast_apply(_DepthwiseConv2d)


def depthwise_conv2d(x, weight, stride=1, pad=(0, 0, 0, 0), bias=None):
    """Depthwise conv (groups == channels) through the HIP kernels.  Under autocast the activations
    run in the autocast dtype (bf16); the weights (and the bias) are consumed in float32 either way."""
    if torch.is_autocast_enabled():
        x = x.to(torch.get_autocast_dtype('cuda'))
    if bias is None:
        return _DW_APPLY(x, weight, int(stride), tuple(int(p) for p in pad))
    return _DW_APPLY(x, weight, int(stride), tuple(int(p) for p in pad), bias)


def depthwise_supported(x, weight, stride):
    """What stp3_dwconv2d_* take: 3x3 / 5x5 at strides 1 and 2, 7x7 at stride 1; channels a multiple of the 16-byte vector."""
    k, s = weight.shape[-1], _pair(stride)
    if not (x.is_cuda and x.dim() == 4 and weight.shape[1] == 1 and weight.shape[-2] == k and s[0] == s[1]):
        return False
    if not ((k in (3, 5) and s[0] in (1, 2)) or (k == 7 and s[0] == 1)):
        return False
    # the kernels compute in bf16 / float32: float32 tensors under a bf16 autocast are cast by the operator, anything else
    # (float64 truth runs, fp16) keeps torch's convolution
    if not (x.dtype in (torch.bfloat16, torch.float32) or
            (torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16)):
        return False
    if torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') != torch.bfloat16:
        return False
    return x.shape[1] % 8 == 0


# ``num_batches_tracked`` -- one int64 increment KERNEL per BatchNorm layer and step in the reference -- is counted on
# the host and applied to all layers with one multi-tensor add: at the end of the optimizer step (parallel.FlatAdam),
# before any ``state_dict()`` of a BatchNorm module, or on ``flush_batch_counters()``.  The counter only feeds the
# cumulative-average mode (momentum=None), which keeps the immediate increment.  Validated bit-identical
# (tests/test_host_cpu.py); measured neutral-to-positive on the MI355X (profiles/r02_ab_greedy_switches.json).
LAZY_COUNTERS = True
_PENDING_COUNTS = {}          # id(buffer) -> [buffer, increments]


def _count_later(bn):
    t = bn.num_batches_tracked
    ent = _PENDING_COUNTS.get(id(t))
    if ent is None or ent[0] is not t:
        _PENDING_COUNTS[id(t)] = [t, 1]
        if not getattr(bn, '_stp3_counter_hook', False):
            bn.register_state_dict_pre_hook(lambda module, prefix, keep_vars: flush_batch_counters())
            bn._stp3_counter_hook = True
    else:
        ent[1] += 1


def pending_batch_counters():
    """The ``num_batches_tracked`` buffers that currently hold un-flushed increments (what one step counted)."""
    return [t for t, _ in _PENDING_COUNTS.values()]


def discard_pending_batch_counters():
    """Forget the increments counted since the last flush: a step that was CAPTURED into a hipGraph counted its BatchNorm
    layers on the host while none of its kernels ran."""
    _PENDING_COUNTS.clear()


def count_batches_again(buffers):
    """One more training-mode forward for every buffer in ``buffers``: a replayed hipGraph of the step runs the BatchNorm
    kernels without passing through ``bump_batch_counter`` (stp3_amd/graph.py)."""
    for t in buffers:
        ent = _PENDING_COUNTS.get(id(t))
        if ent is None or ent[0] is not t:
            _PENDING_COUNTS[id(t)] = [t, 1]
        else:
            ent[1] += 1


def flush_batch_counters():
    """Apply the increments counted since the last flush (no-op when there are none)."""
    if not _PENDING_COUNTS:
        return
    by_count = {}
    for t, k in _PENDING_COUNTS.values():
        by_count.setdefault(k, []).append(t)
    _PENDING_COUNTS.clear()
    with torch.no_grad():
        for k, tensors in by_count.items():
            torch._foreach_add_(tensors, k)


# Set while a block's forward is RE-run during backward (activation recomputation, models/encoder.py): the second run must
# leave the BatchNorm running statistics and batch counters alone -- they were updated by the first.
RECOMPUTING = [False]


class recomputing:
    def __enter__(self):
        self.prev = RECOMPUTING[0]
        RECOMPUTING[0] = True

    def __exit__(self, *exc):
        RECOMPUTING[0] = self.prev
        return False


def bn_momentum(bn):
    """The momentum the kernels update the running statistics with: the module's, or 0 (running = 1 * running + 0 * batch:
    unchanged, bit for bit) while a forward is being recomputed."""
    if RECOMPUTING[0]:
        return 0.0
    return float(bn.momentum if bn.momentum is not None else 0.1)


def bump_batch_counter(bn):
    """``bn.num_batches_tracked += 1`` for a training-mode forward of ``bn`` (immediately, or counted for the flush)."""
    if RECOMPUTING[0]:
        return
    if LAZY_COUNTERS and bn.momentum is not None:
        _count_later(bn)
    else:
        bn.num_batches_tracked.add_(1)


# ----------------------------------------------------------------------------------------------
# fused BatchNorm (+ per-sample bias) + activation (+ residual), cross-replica statistics
# ----------------------------------------------------------------------------------------------
ACT_NONE, ACT_RELU, ACT_SWISH = _lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_SWISH
RES_NONE, RES_BEFORE_ACT, RES_AFTER_ACT = _lib.RES_NONE, _lib.RES_BEFORE_ACT, _lib.RES_AFTER_ACT

_BN_WORKSPACE = {}
_BN_MAX_ROW_BLOCKS = 128          # STP3_BN_MAX_ROW_BLOCKS


def _bn_workspace(n, c, device):
    """Scratch of the two-stage reductions: N * 128 * 3 * C floats at most (one buffer per device, grown on demand)."""
    need = n * _BN_MAX_ROW_BLOCKS * 3 * c * 4
    key = _ws_key(device)
    ws = _BN_WORKSPACE.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 8 << 20), dtype=torch.uint8, device=device)
        _BN_WORKSPACE[key] = ws
    return ws, need


def _rows_view(t):
    """(N, C, H, W) tensor -> (tensor whose memory is [N][H*W][ld] channels-last, ld).  Channels-last tensors
    and channel-slices of them are used in place; anything else is copied to channels-last once."""
    n, c, h, w = t.shape
    if t.is_contiguous(memory_format=torch.channels_last):
        return t, c
    sn, sc, sh, sw = t.stride()
    if c > 1 and sc != 1:
        ok = False
    else:
        ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
        ok = ld >= c and (w == 1 or sw == ld) and (h == 1 or sh == w * ld) and (n == 1 or sn == h * w * ld)
    if not ok:
        t = t.contiguous(memory_format=torch.channels_last)
        ld = c
    return t, ld


def _opt_ptr(t):
    return t.data_ptr() if t is not None else None


def _f32(t):
    if t is None or (t.dtype == torch.float32 and t.is_contiguous()):
        return t
    return t.detach().float().contiguous()


_EXCHANGES = {'batchnorm': 0}


# Tests only: take the cross-replica code path -- split statistics / apply passes with an all-reduce between them, gradient
# buckets reduced from their hooks -- in a process group of ONE rank.  RCCL refuses two ranks on one device
# (scripts/probe_nccl_one_gpu.py), so this is how the captured N > 1 step (stp3_amd/graph.py) is exercised on RCCL on the
# one-GPU boxes the tests run on: every collective is issued (and captured), it just has nobody to talk to.
FORCE_EXCHANGE = False


def replicas(group):
    """(replicas that share their BatchNorm statistics / gradients, whether exchanges are issued) for process group ``group``
    (None: the default group; False: this layer keeps local statistics)."""
    dist = torch.distributed
    if group is False or not (dist.is_available() and dist.is_initialized()):
        return 1, False
    world = dist.get_world_size(group)
    return world, world > 1 or FORCE_EXCHANGE


def exchange_counts(reset=False):
    """How many cross-replica statistics all-reduces the BatchNorm operators have issued (bench.py's N > 1 line)."""
    out = dict(_EXCHANGES)
    if reset:
        for k in _EXCHANGES:
            _EXCHANGES[k] = 0
    return out


def drive_exchange(gen, group):
    """Run an operator written as a generator that YIELDS the buffer it needs summed over the replicas (cross-replica
    BatchNorm statistics; at most one exchange per pass) and resumes once it is: here with one all-reduce of its own."""
    try:
        buf = next(gen)
        while True:
            torch.distributed.all_reduce(buf, group=group)
            _EXCHANGES['batchnorm'] += 1
            buf = gen.send(None)
    except StopIteration as done:
        return done.value


def drive_exchange_group(gens, group):
    """The same for SIBLING operators (parallel branches reading one tensor: the exchange of one does not depend on the
    result of another): all of them run up to their exchange point, ONE all-reduce carries every buffer, all resume."""
    waiting, results = [], [None] * len(gens)
    for i, gen in enumerate(gens):
        try:
            waiting.append((i, gen, next(gen)))
        except StopIteration as done:
            results[i] = done.value
    if waiting:
        _EXCHANGES['batchnorm'] += 1
        if len(waiting) == 1:
            torch.distributed.all_reduce(waiting[0][2], group=group)
        else:
            packed = torch.cat([buf.reshape(-1) for _, _, buf in waiting])
            torch.distributed.all_reduce(packed, group=group)
            for (_, _, buf), part in zip(waiting, packed.split([b.numel() for _, _, b in waiting])):
                buf.copy_(part.view(buf.shape))
        for i, gen, _ in waiting:
            try:
                gen.send(None)
                raise RuntimeError('an operator asked for a second exchange in one pass')
            except StopIteration as done:
                results[i] = done.value
    return results


class _BnAct(torch.autograd.Function):
    """y = act(BN(x + sbias) [+ res]) * oscale [+ res] on (N, C, H, W) tensors, channels-last memory."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, sbias, oscale, running_mean, running_var, training, momentum, eps,
                act, res_mode, group, channels=None, out_slot=None):
        return drive_exchange(_BnAct.forward_steps(ctx, x, weight, bias, res, sbias, oscale, running_mean, running_var,
                                                   training, momentum, eps, act, res_mode, group, channels, out_slot), group)

    @staticmethod
    def backward(ctx, dy):
        return drive_exchange(_BnAct.backward_steps(ctx, dy), ctx.group)

    @staticmethod
    def forward_steps(ctx, x, weight, bias, res, sbias, oscale, running_mean, running_var, training, momentum, eps,
                      act, res_mode, group, channels=None, out_slot=None):
        """``forward`` as a generator: yields the [2C] statistics buffer when it has to be summed over the replicas.
        ``out_slot`` = (buffer, first channel): the result is written into that channel slice of a wider channels-last
        buffer (``ops_fused.join_slices`` then IS the concatenation of the layers that filled it: no copy)."""
        _need_gpu(x)
        if x.dtype == torch.bfloat16:
            dt = _lib.DTYPE_BF16
        elif x.dtype == torch.float32:
            dt = _lib.DTYPE_F32
        else:
            raise _lib.Stp3HipError(f'bn_act supports float32 / bfloat16, got {x.dtype}')
        n, cx, h, w = x.shape
        c = cx if channels is None else int(channels)
        # zero-padded rows (stp3_bn_dims.cpad): x carries cx = pad8(c) lanes per row, the last cx - c of them padding
        # that reads as zero; y / dx get zeros there.  The MFMA convolutions on either side take the rows as they are.
        cpad = 0
        if cx != c:
            if cx != (c + 7) // 8 * 8:
                raise _lib.Stp3HipError(f'bn_act: {cx} channel lanes for {c} channels (want {(c + 7) // 8 * 8})')
            cpad = cx
        dev = x.device
        x, ldx = _rows_view(x)
        ldr = cx
        if res is not None:
            if res.shape != x.shape:
                raise _lib.Stp3HipError('bn_act: residual shape mismatch')
            res, ldr = _rows_view(res if res.dtype == x.dtype else res.to(x.dtype))
        else:
            res_mode = RES_NONE
        if out_slot is None:
            y, ldy = torch.empty((n, cx, h, w), dtype=x.dtype, device=dev, memory_format=torch.channels_last), cx
        else:
            y, ldy = slot_view(out_slot, x)
        dims = _lib.BnDims(n, h * w, c, ldx, ldy, ldr, dt, act, res_mode, sbias is not None, oscale is not None, cpad)
        lib = _lib.lib()
        gamma, beta = _f32(weight), _f32(bias)
        sb, osc = _f32(sbias), _f32(oscale)
        stream = _stream_handle()
        world, exchange = 1, False
        stat = None
        if training:
            ws, ws_bytes = _bn_workspace(n, c, dev)
            stat = torch.empty(4 * c, dtype=torch.float32, device=dev)       # sum | sum of squares | mean | invstd
            base = stat.data_ptr()
            world, exchange = replicas(group)
            if not exchange:
                check(lib.stp3_bn_fwd_train(ctypes.byref(dims), x.data_ptr(), _opt_ptr(sb), _opt_ptr(res), _opt_ptr(osc),
                                            _opt_ptr(gamma), _opt_ptr(beta), eps, momentum, _opt_ptr(running_mean),
                                            _opt_ptr(running_var), base, ws.data_ptr(), ws_bytes, y.data_ptr(), stream),
                      'stp3_bn_fwd_train')
                count = float(n * h * w)
            else:
                check(lib.stp3_bn_stats(ctypes.byref(dims), x.data_ptr(), _opt_ptr(sb), ws.data_ptr(), ws_bytes, base,
                                        stream), 'stp3_bn_stats')
                # cross-replica statistics (train.py:47 sync_batchnorm): one small all-reduce per layer -- or one for
                # several sibling layers (drive_exchange_group); every rank holds the same number of elements
                yield stat[:2 * c]
                count = float(n * h * w) * world
                check(lib.stp3_bn_apply_fwd(ctypes.byref(dims), x.data_ptr(), _opt_ptr(sb), _opt_ptr(res), _opt_ptr(osc),
                                            base, count, _opt_ptr(gamma), _opt_ptr(beta), eps, momentum,
                                            _opt_ptr(running_mean), _opt_ptr(running_var), base + 8 * c, base + 12 * c,
                                            y.data_ptr(), stream), 'stp3_bn_apply_fwd')
        else:
            count = 0.0
            stat = torch.cat([running_mean.detach().float(), running_var.detach().float(), running_mean.detach().float(),
                              torch.rsqrt(running_var.detach().float() + eps)])
            check(lib.stp3_bn_apply_fwd(ctypes.byref(dims), x.data_ptr(), _opt_ptr(sb), _opt_ptr(res), _opt_ptr(osc),
                                        None, 0.0, _opt_ptr(gamma), _opt_ptr(beta), eps, 0.0, running_mean.data_ptr(),
                                        running_var.data_ptr(), None, None, y.data_ptr(), stream), 'stp3_bn_apply_fwd')
        ctx.save_for_backward(x, res if res_mode == RES_BEFORE_ACT else None, sb, osc, gamma, beta, stat)
        ctx.dims, ctx.training, ctx.count, ctx.exchange, ctx.group = dims, training, count, exchange, group
        ctx.res_dtype = None if res is None else res.dtype
        ctx.has_affine = (weight is not None, bias is not None)
        ctx.in_dtypes = (None if weight is None else weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward_steps(ctx, dy):
        x, res, sb, osc, gamma, beta, stat = ctx.saved_tensors
        dims = ctx.dims
        n, rows, c = dims.N, dims.rows, dims.C
        dev = x.device
        lib = _lib.lib()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dy, ldy = _rows_view(dy)
        cx = x.shape[1]                                    # channel lanes per row (> c for zero-padded rows)
        if ldy != cx:
            if ldy % 8 == 0 and dy.data_ptr() % 16 == 0:
                # a channel slice of a wider channels-last tensor (the gradient of a torch.cat): read in place with its
                # own row stride instead of being copied dense first
                dims = _lib.BnDims(dims.N, dims.rows, dims.C, dims.ldx, ldy, dims.ldr, dims.dtype, dims.act,
                                   dims.res_mode, dims.has_sbias, dims.has_oscale, dims.cpad)
            else:
                dy = dy.contiguous(memory_format=torch.channels_last)
        ws, ws_bytes = _bn_workspace(n, c, dev)
        sumbuf = torch.empty((n + 1) * 3 * c, dtype=torch.float32, device=dev)   # [N][3][C] per sample | [3][C] total
        sums_off = n * 3 * c
        mean_p, invstd_p = stat.data_ptr() + 8 * c, stat.data_ptr() + 12 * c
        stream = _stream_handle()
        if x.is_contiguous(memory_format=torch.channels_last):
            dx = torch.empty_like(x)
        else:
            dx = torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=dev)   # channel-sliced view of a wider tensor
        dres = None
        bdims = dims
        if dims.res_mode == RES_BEFORE_ACT and ctx.needs_input_grad[3]:
            dres = torch.empty((n, cx) + tuple(x.shape[2:]), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        simple = ctx.training and not ctx.exchange and (dres is None or dims.ldr == cx)
        if simple:
            check(lib.stp3_bn_bwd_train(ctypes.byref(dims), dy.data_ptr(), x.data_ptr(), _opt_ptr(sb), _opt_ptr(res),
                                        _opt_ptr(osc), mean_p, invstd_p, _opt_ptr(gamma), _opt_ptr(beta), ws.data_ptr(),
                                        ws_bytes, sumbuf.data_ptr(), dx.data_ptr(), _opt_ptr(dres), stream),
                  'stp3_bn_bwd_train')
            gsums = sumbuf[sums_off:].view(3, c)
            lsums = gsums
        else:
            check(lib.stp3_bn_bwd_reduce(ctypes.byref(dims), dy.data_ptr(), x.data_ptr(), _opt_ptr(sb), _opt_ptr(res),
                                         _opt_ptr(osc), mean_p, invstd_p, _opt_ptr(gamma), _opt_ptr(beta), ws.data_ptr(),
                                         ws_bytes, sumbuf.data_ptr(), sumbuf.data_ptr() + 4 * sums_off, stream),
                  'stp3_bn_bwd_reduce')
            lsums = sumbuf[sums_off:].view(3, c)
            gsums = lsums
            if ctx.training and ctx.exchange:
                gsums = lsums.clone()
                yield gsums
            if dres is not None and dims.ldr != cx:
                # the reduce pass read `res` with its own stride; the apply pass writes dres densely and reads res
                # with the same stride
                res = res.contiguous(memory_format=torch.channels_last)
                bdims = _lib.BnDims(dims.N, dims.rows, dims.C, dims.ldx, dims.ldy, cx, dims.dtype, dims.act,
                                    dims.res_mode, dims.has_sbias, dims.has_oscale, dims.cpad)
            check(lib.stp3_bn_apply_bwd(ctypes.byref(bdims), dy.data_ptr(), x.data_ptr(), _opt_ptr(sb), _opt_ptr(res),
                                        _opt_ptr(osc), mean_p, invstd_p, _opt_ptr(gamma), _opt_ptr(beta),
                                        gsums.data_ptr() if ctx.training else None, max(ctx.count, 1.0), dx.data_ptr(),
                                        _opt_ptr(dres), stream), 'stp3_bn_apply_bwd')
        dgamma = dbeta = None
        if ctx.has_affine[0] and ctx.needs_input_grad[1]:
            dgamma = lsums[1] if ctx.in_dtypes[0] == torch.float32 else lsums[1].to(ctx.in_dtypes[0])
        if ctx.has_affine[1] and ctx.needs_input_grad[2]:
            dbeta = lsums[0] if ctx.in_dtypes[1] == torch.float32 else lsums[0].to(ctx.in_dtypes[1])
        if dims.res_mode == RES_AFTER_ACT and ctx.needs_input_grad[3]:
            dres = dy
        if dres is not None and ctx.res_dtype is not None and dres.dtype != ctx.res_dtype:
            dres = dres.to(ctx.res_dtype)
        dsbias = None
        if sb is not None and ctx.needs_input_grad[4]:
            # one launch (stp3_bn_dsbias) instead of six small torch operators per layer
            dsbias = torch.empty(n, c, dtype=torch.float32, device=dev)
            check(lib.stp3_bn_dsbias(n, c, rows, sumbuf.data_ptr(), gsums.data_ptr() if ctx.training else None,
                                     max(ctx.count, 1.0), _opt_ptr(gamma), stat.data_ptr() + 12 * c, dsbias.data_ptr(), stream),
                  'stp3_bn_dsbias')
        return dx, dgamma, dbeta, dres, dsbias, None, None, None, None, None, None, None, None, None, None, None


_BN_APPLY = _fast_apply(_BnAct)


def slot_view(out_slot, like):
    """(channel-slice view of the slot's buffer for a result shaped like ``like``, row stride of the buffer)."""
    buf, c0 = out_slot
    n, c, h, w = like.shape
    if not (buf.is_contiguous(memory_format=torch.channels_last) and buf.shape[0] == n and tuple(buf.shape[2:]) == (h, w)
            and buf.dtype == like.dtype and c0 % 8 == 0 and c0 + c <= buf.shape[1]):
        raise _lib.Stp3HipError('output slot does not fit the result (shape / dtype / channels-last / 16-byte channel offset)')
    return buf[:, c0:c0 + c], buf.shape[1]


def bn_act(x, weight, bias, running_mean, running_var, training, momentum, eps, act=ACT_NONE, res=None,
           res_mode=RES_NONE, sbias=None, oscale=None, group=None, channels=None, out_slot=None):
    """Fused BatchNorm + activation (+ residual) through the HIP kernels (GPU tensors only).
    ``group=False`` disables the cross-replica statistics even when torch.distributed is initialised.
    ``channels``: the BatchNorm's channel count when x (and res) carry rows zero-padded to a multiple of 8 lanes
    (``x.shape[1] == pad8(channels)``); the result has the same padded shape with zeros in the extra lanes."""
    if res is None:
        res_mode = RES_NONE
    return _BN_APPLY(x, weight, bias, res, sbias, oscale, running_mean, running_var, bool(training),
                        float(momentum if momentum is not None else 0.1), float(eps), int(act), int(res_mode), group,
                        channels, out_slot)


class _FanOut(torch.autograd.Function):
    """x -> n aliases of x, one per consumer.  What this changes is the BACKWARD: autograd adds the n gradients arriving at
    a tensor pairwise (n - 1 launches, each reading two tensors, writing one and rounding to bf16); here they are added
    in ONE pass (stp3_sum_n: float32 accumulation in consumer order, one rounding)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        # a handle nobody consumed hands back None, not a zero tensor (which would be filled, and -- contiguous where the real
        # gradients are channels-last -- push the whole sum onto strided pairwise additions: 5 x 73 us in the decoder)
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        # gradients that are constant over every (sample, channel) plane -- a whole-plane mean hands its gradient back as a
        # stride-0 view of an (N, C) tensor (layers/fused._PlaneMean) -- join the sum as ONE broadcast addend instead of
        # being written out as full tensors and read back (41 us + a fifth of the sum's reads per 61 MB tensor)
        planes = [g for g in gs if _is_plane_constant(g)]
        dense = [g for g in gs if not _is_plane_constant(g)]
        if planes and dense:
            first = dense[0]
            n_, c_, h_, w_ = first.shape
            per = 8 if first.dtype == torch.bfloat16 else 4
            same = all(g.shape == first.shape and g.dtype == first.dtype and g.stride() == first.stride() for g in dense)
            if (first.is_cuda and same and first.dtype in (torch.float32, torch.bfloat16) and len(dense) <= 8 and c_ % per == 0
                    and first.is_contiguous(memory_format=torch.channels_last) and all(p.shape == first.shape for p in planes)):
                plane = planes[0][:, :, 0, 0]
                for p in planes[1:]:
                    plane = plane + p[:, :, 0, 0]
                plane = plane.to(first.dtype).contiguous()
                out = torch.empty_like(first)
                arr = (ctypes.c_void_p * len(dense))(*[g.data_ptr() for g in dense])
                check(_lib.lib().stp3_sum_n_plane(len(dense), first.numel(), _lib.DTYPE_BF16 if first.dtype == torch.bfloat16
                                                  else _lib.DTYPE_F32, arr, plane.data_ptr(), c_ * h_ * w_, c_, out.data_ptr(),
                                                  _stream_handle()), 'stp3_sum_n_plane')
                return out, None
        first = gs[0]
        same = all(g.shape == first.shape and g.dtype == first.dtype and g.stride() == first.stride() for g in gs)
        if (first.is_cuda and same and first.dtype in (torch.float32, torch.bfloat16) and len(gs) <= 8
                and _dense_layout(first)):
            out = torch.empty_like(first)
            arr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
            check(_lib.lib().stp3_sum_n(len(gs), first.numel(), _lib.DTYPE_BF16 if first.dtype == torch.bfloat16 else _lib.DTYPE_F32,
                                        arr, out.data_ptr(), _stream_handle()), 'stp3_sum_n')
            return out, None
        total = gs[0]
        for g in gs[1:]:
            total = total + g
        return total, None


def _is_plane_constant(g):
    return g.dim() == 4 and g.stride(2) == 0 and g.stride(3) == 0 and g.shape[2] * g.shape[3] > 1


def _dense_layout(t):
    """Non-overlapping and dense (any permutation of a contiguous layout): element order in memory is well defined."""
    expect = 1
    for st, sz in sorted((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1):
        if st != expect:
            return False
        expect *= sz
    return True


def fan_out(x, n):
    """n handles on x for n consumers (parallel branches reading one tensor); their gradients are added in one pass."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return [x] * max(n, 1)
    return list(_FanOut.apply(x, n))


def _rows_f32(w):
    """A float32 matrix as (tensor, row stride) for the kernels: rows of unit stride, any row stride (the columns of a wider
    parameter, read in place); anything else is copied."""
    if w.stride(1) != 1 or w.stride(0) < w.shape[1]:
        w = w.contiguous()
    return w, w.stride(0)


_COLUMN_DESTS = {}        # (address, rows, columns) of a ``weight_columns`` tensor -> (entry, parameter, c0, c1)


class _SmallLinear(torch.autograd.Function):
    """y = x W^T + b for a handful of rows, float32 (stp3_linear_fwd / _bwd): one launch each way.  W may be a run of columns
    of a wider parameter (``weight_columns``): it is read in place, and its gradient is written straight into the same columns
    of the parameter's bucket slice when the parameter takes direct gradients (``_direct_ok``)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_gpu(x, w)
        x = x.contiguous()
        w, ldw = _rows_f32(w)
        m, k = x.shape
        n = w.shape[0]
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        check(_lib.lib().stp3_linear_fwd(m, k, n, x.data_ptr(), w.data_ptr(), ldw, _opt_ptr(b), y.data_ptr(), _stream_handle()),
              'stp3_linear_fwd')
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        m, k = x.shape
        n = w.shape[0]
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dest = _COLUMN_DESTS.get((w.data_ptr(), n, k))
            if dest is not None and dest[1]() is not None and _direct_ok(dest[0], _graph_task_id(), dy.device, True):
                view = dest[1]()._stp3_grad_view        # the columns of the parameter's bucket slice
                dw = view.as_strided((n, k), (view.stride(0), view.stride(1)), view.storage_offset() + dest[2] * view.stride(1))
            else:
                dw = torch.empty((n, k), dtype=torch.float32, device=x.device)
        db = torch.empty(n, dtype=torch.float32, device=x.device) if ctx.has_bias and ctx.needs_input_grad[2] else None
        check(_lib.lib().stp3_linear_bwd(m, k, n, dy.data_ptr(), x.data_ptr(), w.data_ptr(), w.stride(0), _opt_ptr(dx), _opt_ptr(dw),
                                         0 if dw is None else dw.stride(0), _opt_ptr(db), _stream_handle()), 'stp3_linear_bwd')
        return dx, dw, db


def small_linear_supported(x, w, b=None):
    """float32 GPU matrices, few enough multiply-adds that the product is launch latency (the pooled descriptors: <= 72 rows)."""
    return (x.is_cuda and x.dim() == 2 and w.dim() == 2 and x.dtype == w.dtype == torch.float32 and x.shape[1] == w.shape[1]
            and (b is None or (b.dtype == torch.float32 and b.dim() == 1 and b.is_contiguous()))
            and x.shape[0] * x.shape[1] * w.shape[0] <= (1 << 22))


def small_linear(x, w, b=None):
    return _SmallLinear.apply(x, w, b)


class _UpsampleBilinear(torch.autograd.Function):
    """Bilinear up-sampling by an integer factor, align_corners=False (stp3_upsample_bilinear_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, scale, out_slot=None):
        _need_gpu(x)
        n, c, h, w = x.shape
        x, ldx = _rows_view(x)
        dt = _lib.DTYPE_BF16 if x.dtype == torch.bfloat16 else _lib.DTYPE_F32
        if out_slot is None:
            y, ldy = torch.empty((n, c, h * scale, w * scale), dtype=x.dtype, device=x.device, memory_format=torch.channels_last), c
        else:
            # written straight into its channel slice of a concatenation's buffer (ops_fused.join_slices)
            y, ldy = slot_view(out_slot, torch.empty((n, c, h * scale, w * scale), dtype=x.dtype, device='meta'))
        dims = _lib.UpsampleDims(n, h, w, c, scale, ldx, ldy, dt)
        check(_lib.lib().stp3_upsample_bilinear_fwd(ctypes.byref(dims), x.data_ptr(), y.data_ptr(), _stream_handle()),
              'stp3_upsample_bilinear_fwd')
        ctx.shape = (n, c, h, w, scale, dt)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, scale, dt = ctx.shape
        want = torch.bfloat16 if dt == _lib.DTYPE_BF16 else torch.float32
        if dy.dtype != want:
            dy = dy.to(want)
        dy, ldy = _rows_view(dy)
        if ldy % (8 if dt == _lib.DTYPE_BF16 else 4) != 0 or dy.data_ptr() % 16 != 0:
            dy, ldy = dy.contiguous(memory_format=torch.channels_last), c
        dx = torch.empty((n, c, h, w), dtype=want, device=dy.device, memory_format=torch.channels_last)
        dims = _lib.UpsampleDims(n, h, w, c, scale, c, ldy, dt)
        check(_lib.lib().stp3_upsample_bilinear_bwd(ctypes.byref(dims), dy.data_ptr(), dx.data_ptr(), _stream_handle()),
              'stp3_upsample_bilinear_bwd')
        return dx, None, None


def upsample_bilinear_supported(x, scale):
    """stp3_upsample_bilinear_fwd: GPU bf16 / float32 (N, C, H, W), C in whole 16-byte vectors, integer scale 1..4."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32)):
        return False
    per = 8 if x.dtype == torch.bfloat16 else 4
    return float(scale) == int(scale) and 1 <= int(scale) <= 4 and x.shape[1] % per == 0


def upsample_bilinear(x, scale, out_slot=None):
    """F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=False) for channels-last GPU tensors, in the
    tensor's own dtype (float32 arithmetic, one rounding); a channel slice of a concatenation's gradient is read in
    place by the backward."""
    return _UpsampleBilinear.apply(x, int(scale), out_slot)


class _CausalPair(torch.autograd.Function):
    """[x[t-1], x[t]] channel pairing of the frame-folded sequence (stp3_causal_pair_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, frames_per_sample):
        _need_gpu(x)
        n, c, h, w = x.shape
        x, ldx = _rows_view(x)
        dt = _lib.DTYPE_BF16 if x.dtype == torch.bfloat16 else _lib.DTYPE_F32
        dims = _lib.PairDims(n, int(frames_per_sample), h * w, c, ldx, dt)
        y = torch.empty((n, 2 * c, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        check(_lib.lib().stp3_causal_pair_fwd(ctypes.byref(dims), x.data_ptr(), y.data_ptr(), _stream_handle()),
              'stp3_causal_pair_fwd')
        ctx.dims = dims
        return y

    @staticmethod
    def backward(ctx, dy):
        d = ctx.dims
        dy = dy.contiguous(memory_format=torch.channels_last)
        n, c2, h, w = dy.shape
        dx = torch.empty((n, c2 // 2, h, w), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        dims = _lib.PairDims(d.frames, d.T, d.rows, d.C, d.C,
                             _lib.DTYPE_BF16 if dy.dtype == torch.bfloat16 else _lib.DTYPE_F32)
        check(_lib.lib().stp3_causal_pair_bwd(ctypes.byref(dims), dy.data_ptr(), dx.data_ptr(), _stream_handle()),
              'stp3_causal_pair_bwd')
        return dx, None


def causal_pair_supported(x):
    """stp3_causal_pair_fwd takes GPU bf16 / float32 (N, C, H, W) tensors whose C channels fill 16-byte vectors."""
    per = 8 if x.dtype == torch.bfloat16 else 4
    return x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and x.shape[1] % per == 0


def causal_pair(x, frames_per_sample):
    """x (B*T, C, H, W), the T frames of a sample consecutive -> (B*T, 2C, H, W) = [previous frame (zero for the first
    of each sample), this frame] along the channels: the operand of the causal (2,3,3) convolution."""
    return _CausalPair.apply(x, frames_per_sample)


# ----------------------------------------------------------------------------------------------
# dense 2-D convolution: bf16 MFMA implicit GEMM (forward and stride-1 data gradient)
# ----------------------------------------------------------------------------------------------
def _pair(v):
    return (int(v), int(v)) if not isinstance(v, (tuple, list)) else (int(v[0]), int(v[1]))


def _conv_out(size, k, stride, pad, dil):
    return (size + 2 * pad - dil * (k - 1) - 1) // stride + 1


_CONV_STAT_WS = {}


class SkipCarrier:
    """The gradient of a block's identity skip, handed from the operator that adds the skip (the BatchNorm at the block's end:
    its backward runs first and stores the gradient here instead of returning it) to the operator that consumes the block
    input (the block's first convolution: its data-gradient kernel adds it in its epilogue, stp3_conv2d_fwd_add).  Replaces the
    pass torch.autograd makes over both tensors when a tensor has two consumers.  One carrier per block and forward pass; both
    operators must belong to the same backward pass (they do: one feeds the other)."""
    __slots__ = ('grad',)

    def __init__(self):
        self.grad = None

    def take(self):
        g, self.grad = self.grad, None
        return g


def _conv2d_launch(x, wb, bias, stride, pad, dil, out_dtype, sums_ptr=None, out_hw=None, add=None):
    """x (N,Cin,H,W) bf16 with channels-last memory (row stride ld >= Cin); wb (Cout,Cin,KH,KW) bf16 channels-last.
    ``sums_ptr``: device address of a float32 [2][Cout] buffer that receives the BatchNorm statistics of y (bf16 y).
    ``out_hw``: output size when it is not the one ``pad`` implies on both sides (``pad`` is the top / left padding;
    taps that fall off the bottom / right edge read zeros like any other padding)."""
    n, cin, h, w = x.shape
    cout, _, kh, kw = wb.shape
    x, ldx = _rows_view(x)
    if out_hw is None:
        ho, wo = _conv_out(h, kh, stride, pad[0], dil[0]), _conv_out(w, kw, stride, pad[1], dil[1])
    else:
        ho, wo = out_hw
    y = torch.empty((n, cout, ho, wo), dtype=out_dtype, device=x.device, memory_format=torch.channels_last)
    dims = _lib.ConvDims(n, h, w, cin, ho, wo, cout, kh, kw, stride, pad[0], pad[1], dil[0], dil[1], ldx, cout,
                         _lib.DTYPE_F32 if out_dtype == torch.float32 else _lib.DTYPE_BF16, int(bias is not None))
    lib = _lib.lib()
    ws_ptr, ws_bytes = None, 0
    if sums_ptr is not None:
        nbytes = ctypes.c_size_t()
        check(lib.stp3_conv2d_fwd_workspace(ctypes.byref(dims), ctypes.byref(nbytes)), 'stp3_conv2d_fwd_workspace')
        key = _ws_key(x.device)
        ws = _CONV_STAT_WS.get(key)
        if ws is None or ws.numel() < nbytes.value:
            ws = torch.empty(max(nbytes.value, 8 << 20), dtype=torch.uint8, device=x.device)
            _CONV_STAT_WS[key] = ws
        ws_ptr, ws_bytes = ws.data_ptr(), nbytes.value
    if add is not None:
        # y = conv + add in the kernel's epilogue (the skip's gradient joins the data gradient where it is written)
        assert sums_ptr is None
        add, ldadd = _rows_view(add)
        if (out_dtype == torch.bfloat16 and add.dtype == torch.bfloat16 and tuple(add.shape) == tuple(y.shape) and cout % 8 == 0
                and ldadd % 8 == 0 and add.data_ptr() % 16 == 0):
            check(lib.stp3_conv2d_fwd_add(ctypes.byref(dims), _ptr(x), _ptr(wb), _opt_ptr(bias), _ptr(add), ldadd, _ptr(y),
                                          _stream()), 'stp3_conv2d_fwd_add')
            return y
        check(lib.stp3_conv2d_fwd(ctypes.byref(dims), _ptr(x), _ptr(wb), _opt_ptr(bias), _ptr(y), None, None, 0, _stream()),
              'stp3_conv2d_fwd')
        return y + add.to(y.dtype)
    check(lib.stp3_conv2d_fwd(ctypes.byref(dims), _ptr(x), _ptr(wb), _opt_ptr(bias), _ptr(y), sums_ptr, ws_ptr, ws_bytes,
                              _stream()), 'stp3_conv2d_fwd')
    return y


_CONV_WORKSPACE = {}


# Weight gradients of LEAF parameters are written straight into the parameter's slice of its flat gradient bucket
# (parallel.GradientBuckets), see ``_conv2d_wgrad``.
DIRECT_BUCKET_GRADS = True


def _conv2d_wgrad(dy, x, wshape, stride, pad, dil, leaf=None):
    """dw (Cout,Cin,KH,KW) float32, channels-last memory, through stp3_conv2d_wgrad (bf16 operands).
    ``leaf``: the tensor the operator received as its weight.  When that is a LEAF parameter without a gradient yet, in
    float32 and in the memory order of dw, autograd's AccumulateGrad keeps the tensor handed back as the parameter's ``.grad``
    without launching anything -- so it can be written where the optimizer reads it (below)."""
    cout, cin, kh, kw = wshape
    n, _, h, w = x.shape
    x, ldx = _rows_view(x)
    dy, ldy = _rows_view(dy)
    ho, wo = dy.shape[2], dy.shape[3]
    dims = _lib.ConvDims(n, h, w, cin, ho, wo, cout, kh, kw, stride, pad[0], pad[1], dil[0], dil[1], ldx, ldy,
                         _lib.DTYPE_F32, 0)
    lib = _lib.lib()
    nbytes = ctypes.c_size_t()
    check(lib.stp3_conv2d_wgrad_workspace(ctypes.byref(dims), ctypes.byref(nbytes)), 'stp3_conv2d_wgrad_workspace')
    dw = torch.empty((cout, cin, kh, kw), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    # a LEAF parameter without a gradient yet, float32, in the memory order of dw: AccumulateGrad keeps the tensor we hand back
    # as the parameter's .grad without launching anything
    direct = False
    plain_leaf = (leaf is not None and x.is_cuda and leaf.is_leaf and leaf.grad is None and leaf.dtype == torch.float32
                  and tuple(leaf.shape) == tuple(dw.shape) and _same_memory_order(leaf, dw))
    if plain_leaf:
        # ... so the gradient can be written where the optimizer reads it: the parameter's slice of its flat gradient bucket
        # (parallel.GradientBuckets), a FRESH alias of it (autograd only keeps a tensor nobody else holds) -- the gather copy
        # of the bucket then skips this parameter
        # Once per backward pass and parameter: a weight that is used several times in one graph (the GRU cells of the
        # prediction stage) gets its second and later contributions in fresh tensors, which autograd adds to the first.
        view = getattr(leaf, '_stp3_grad_view', None)
        task = _graph_task_id()
        if (DIRECT_BUCKET_GRADS and view is not None and task != -1 and getattr(leaf, '_stp3_grad_claim', None) != task and view.device == dw.device
                and view.dtype == torch.float32 and _same_memory_order(view, dw)):
            leaf._stp3_grad_claim = task
            dw = view.detach()
            direct = True
    asm = getattr(leaf, '_stp3_assembled', None) if leaf is not None else None
    if (asm is not None and ASSEMBLED_WEIGHTS and DIRECT_BUCKET_GRADS and x.is_cuda and leaf._stp3_assembled_direct
            and _assembled_claim(asm, dw, leaf._stp3_assembled_direct == 'shared')):
        # the stand-in of an assembled weight: the gradient of the whole goes to the entry's buffer, and one launch per backward
        # pass (``_WeightShadows.scatter``) cuts every such buffer into the parameters' bucket slices
        dw = asm['dw'].detach()
        direct = True
    key = _ws_key(x.device)
    if direct and DEFER_WGRAD_REDUCE and getattr(leaf, '_stp3_uses', 0) == 1 and _single_process():
        # nobody reads this gradient before the optimizer: only the split contraction runs now, its partial sums stay in this
        # layer's slice of the arena, and ``flush_wgrad_reductions`` (GradientBuckets.finish) sums all layers' in one launch
        arena = _WGRAD_ARENAS.get(key)
        if arena is None:
            arena = _WGRAD_ARENAS[key] = _WgradArena(x.device)
        slot = arena.take(nbytes.value)
        if slot is not None:
            splits = ctypes.c_int32()
            check(lib.stp3_conv2d_wgrad_partials(ctypes.byref(dims), _ptr(dy), _ptr(x), slot, ctypes.c_size_t(nbytes.value),
                                                 ctypes.byref(splits), _stream()), 'stp3_conv2d_wgrad_partials')
            arena.jobs.append((slot, dw.data_ptr(), cout * cin * kh * kw, splits.value))
            return dw
    ws = _CONV_WORKSPACE.get(key)
    if ws is None or ws.numel() < nbytes.value:
        ws = torch.empty(max(nbytes.value, 64 << 20), dtype=torch.uint8, device=x.device)
        _CONV_WORKSPACE[key] = ws
    check(lib.stp3_conv2d_wgrad(ctypes.byref(dims), _ptr(dy), _ptr(x), _ptr(dw), _ptr(ws), ctypes.c_size_t(nbytes.value),
                                _stream()), 'stp3_conv2d_wgrad')
    return dw


# The split-K reductions of the weight gradients that go straight into their bucket slice are DEFERRED to one launch at the
# end of the backward pass (weights applied exactly once since zero_grad -- ``note_weight_use`` --, single process only: with more ranks the bucket hooks all-reduce a bucket as soon as its last
# gradient lands, so the gradient must be complete when the operator returns).
DEFER_WGRAD_REDUCE = True
_WGRAD_ARENAS = {}


def note_weight_use(weight):
    """Called by the forward of every operator that hands ``leaf=weight`` to ``_conv2d_wgrad``: counts the applications of a
    weight since the last ``GradientBuckets.zero_grad``.  A weight applied more than once in one graph (the GRU cells of the
    prediction stage, a block re-run under activation recomputation) has its contributions ADDED by the autograd engine as
    they arrive -- the first one must be complete by then, so only single-use weights may defer their split-K sum."""
    if weight is not None:
        weight._stp3_uses = getattr(weight, '_stp3_uses', 0) + 1
    return weight


def _single_process():
    return not replicas(None)[1]


class _WgradArena:
    """Partial sums of the deferred weight gradients of one backward pass: a bump allocator over one device buffer.  A pass that
    needs more than the buffer holds takes the immediate path for the layers that do not fit and the buffer grows BETWEEN
    passes (never while partials are live); the same sequence of layers gets the same addresses in every pass, which is what
    a captured step needs."""

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(64 << 20, dtype=torch.uint8, device=device)
        self.retired = []              # outgrown buffers stay allocated: a captured step may still name their addresses
        self.used = 0
        self.wanted = 0
        self.jobs = []

    def take(self, nbytes):
        nbytes = (nbytes + 255) // 256 * 256
        self.wanted += nbytes
        if self.used + nbytes > self.buf.numel():
            return None
        ptr = self.buf.data_ptr() + self.used
        self.used += nbytes
        return ptr

    def flush(self):
        if self.jobs:
            arr = (_lib.WgradJob * len(self.jobs))()
            for rec, (partials, dw, numel, splits) in zip(arr, self.jobs):
                rec.partials, rec.dw, rec.numel, rec.splits = partials, dw, numel, splits
            check(_lib.lib().stp3_conv2d_wgrad_reduce_batch(len(self.jobs), arr, _stream()), 'stp3_conv2d_wgrad_reduce_batch')
        grow = self.wanted > self.buf.numel() and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())
        self.jobs, self.used, wanted, self.wanted = [], 0, self.wanted, 0
        if grow:
            self.retired.append(self.buf)
            self.buf = torch.empty(wanted + (wanted >> 3), dtype=torch.uint8, device=self.device)


def flush_wgrad_reductions():
    """Sum the deferred weight-gradient partials of the backward pass that just ran (one launch per device).  Called by
    ``GradientBuckets.finish``; a no-op when nothing is pending."""
    for arena in _WGRAD_ARENAS.values():
        if arena.jobs or arena.wanted:
            arena.flush()
    for tab in _SHADOW_TABLES.values():             # (after the sums: the assembled weights' gradients are among them)
        if tab.pending_scatter:
            ents, tab.pending_scatter = tab.pending_scatter, []
            tab.scatter(ents)


def pending_wgrad_reductions():
    return sum(len(a.jobs) for a in _WGRAD_ARENAS.values()) + sum(len(t.pending_scatter) for t in _SHADOW_TABLES.values())


def reset_weight_uses():
    """``GradientBuckets.zero_grad``: a new pass -- no assembled weight has been applied in it yet."""
    for tab in _SHADOW_TABLES.values():
        for e in tab.assembled.values():
            e['uses'] = 0
            e['columns'] = []


def _same_memory_order(a, b):
    """Same shape and the same stride in every dimension that has more than one element (a 1 x 1 kernel's channels-last and
    contiguous strides differ only in dimensions of size one: the same memory)."""
    return tuple(a.shape) == tuple(b.shape) and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


def _graph_task_id():
    """Id of the backward pass the autograd engine is running on this thread, -1 outside of one."""
    try:
        return torch._C._current_graph_task_id()
    except AttributeError:
        return -1


_CONV_MAX_STAT_TILES = 65536          # stp3_conv2d_fwd: row tiles of 128 pixels the statistics epilogue can reduce


def conv2d_supported(x, weight, stride, groups=1):
    """What stp3_conv2d_fwd / _wgrad take: GPU, dense (groups == 1), square stride, input channels a multiple of 8, and
    fewer than 2^31 PIXELS (the kernels index pixels with 32 bits -- the weight gradient keeps two images of slack and
    packs rows / columns into 16 bits -- and address elements with 64-bit pointer arithmetic: the 4.6 G-element expanded
    tensors of BASELINE configs[4] at three samples per GPU stay on the kernels)."""
    s = _pair(stride)
    if not (x.is_cuda and x.dim() == 4 and groups == 1 and s[0] == s[1] and weight.shape[1] % 8 == 0):
        return False
    n, c, h, w = x.shape
    return (n + 2) * h * w < (1 << 31) and h < 32768 and w < 32768


def conv2d_stats_supported(x, weight, stride, padding=0, dilation=1):
    """``conv2d_supported`` + the bound of the BatchNorm-statistics epilogue: at most 65 536 row tiles of 128 output
    pixels (8.4 M pixels: batch 17 per GPU at the 112 x 240 trunk layers).  Beyond it the caller takes the
    convolution and the BatchNorm as separate operators (the statistics pass has no such bound)."""
    if not conv2d_supported(x, weight, stride):
        return False
    s, p, d = _pair(stride), _pair(padding), _pair(dilation)
    n, _, h, w = x.shape
    ho = _conv_out(h, weight.shape[2], s[0], p[0], d[0])
    wo = _conv_out(w, weight.shape[3], s[1], p[1], d[1])
    return (n * ho * wo + 127) // 128 <= _CONV_MAX_STAT_TILES


# bf16 copies of the convolution weights (forward layout and the tap-flipped / channel-swapped layout of the
# data gradient), keyed by the parameter and invalidated by its version counter: one cast per optimizer step
# instead of one per use.
_WEIGHT_CACHE = {}
_WEIGHT_EPOCH = [0]


# The bf16 copies of PARAMETERS are persistent shadow buffers that one launch of stp3_conv2d_prep_weights rewrites for
# all layers after an optimizer step, instead of ~5 torch operators per layer (validated on the MI355X, round 2).


class _WeightShadows:
    """The shadows of the parameters of ONE device (the table holds device pointers and the rewrite is one launch
    on that device)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.entries = {}            # id(parameter) -> entry
        self.assembled = {}          # key -> entry of a weight assembled from parameter views (``assembled_weight``)
        self.scatter_tables = {}     # set of assembled weights -> device table of their gradient scatter
        self.pending_scatter = []    # assembled weights whose gradient waits for the end of the backward pass
        self.rows = 0
        self.order = []
        self.table = None            # stp3_wprep_entry[n] in device memory
        self.total_blocks = 0

    def lookup(self, weight):
        ent = self.entries.get(id(weight))
        if ent is None or ent['ref']() is not weight or ent['ptr'] != weight.data_ptr():
            return None
        if ent['version'] != weight._version:          # updated in place behind our back (a torch optimizer)
            self.refresh()
        return ent

    def register(self, weight):
        cout, cin, kh, kw = weight.shape
        opts = dict(dtype=torch.bfloat16, device=weight.device, memory_format=torch.channels_last)
        ent = {'ref': weakref.ref(weight), 'ptr': weight.data_ptr(), 'version': weight._version,
               'wb': torch.empty((cout, cin, kh, kw), **opts),         # memory [Cout][KH][KW][Cin]
               'wt': torch.empty((cin, cout, kh, kw), **opts)}         # memory [Cin][KH][KW][Cout], taps flipped
        self.entries[id(weight)] = ent
        self.order = [e for e in self.order if _shadow_alive(e) and e is not ent] + [ent]
        self._build_table()
        self.refresh()
        return ent

    def register_taps(self, weight):
        """A depthwise (C,1,K,K) parameter: its tap-major float32 copy [K*K][C] (what stp3_dwconv2d_* read) joins the table."""
        c, _, k, _ = weight.shape
        ent = {'ref': weakref.ref(weight), 'ptr': weight.data_ptr(), 'version': weight._version, 'kind': 'taps',
               'taps': torch.empty((k * k, c), dtype=torch.float32, device=weight.device)}
        self.entries[id(weight)] = ent
        self.order = [e for e in self.order if _shadow_alive(e) and e is not ent] + [ent]
        self._build_table()
        self.refresh()
        return ent

    def register_assembled(self, key, shape, pieces):
        """A weight ASSEMBLED from views of parameters (``assembled_weight``): shadows of the whole, one table row per piece."""
        cout, cin, kh, kw = shape
        dev = self.device
        opts = dict(dtype=torch.bfloat16, device=dev)
        ent = {'kind': 'assembled', 'key': key, 'shape': tuple(shape), 'pieces': pieces, 'uses': 0, 'claim': None, 'columns': [],
               'signature': _pieces_signature(shape, pieces),
               # (zeroed ONCE: the lanes between the pieces are never written)
               'wb': torch.zeros((cout, cin, kh, kw), **opts).contiguous(memory_format=torch.channels_last),
               'wt': torch.zeros((cin, cout, kh, kw), **opts).contiguous(memory_format=torch.channels_last),
               # the float32 gradient of the whole, [Cout][KH][KW][Cin]: stp3_conv2d_wgrad writes it, one launch per backward pass
               # cuts it into the parameters' gradients
               'dw': torch.zeros((cout, cin, kh, kw), dtype=torch.float32, device=dev).contiguous(memory_format=torch.channels_last),
               # what the operators receive as "the weight": shape and dtype only -- every kernel reads the shadows; NaN so
               # that anything that did read it shows
               'token': torch.full((1,), float('nan'), dtype=torch.float32, device=dev).expand(cout, cin, kh, kw),
               'scatter_table': None}
        self.assembled[key] = ent
        self.order = [e for e in self.order if _shadow_alive(e) and e is not ent] + [ent]
        self._build_table()
        self.refresh()
        return ent

    def _rows(self, e):
        kind = e.get('kind', 'leaf')
        if kind == 'leaf':
            w = e['ref']()
            return [dict(src=w.data_ptr(), fwd=e['wb'].data_ptr(), flip=e['wt'].data_ptr(), strides=w.stride(), dims=tuple(w.shape))]
        if kind == 'taps':
            w = e['ref']()
            c, _, k, _ = w.shape
            return [dict(src=w.data_ptr(), fwd=e['taps'].data_ptr(), flip=0, strides=(0, w.stride(0), w.stride(2), w.stride(3)),
                         dims=(1, c, k, k), fwd_f32=1)]
        cout, cin = e['shape'][:2]
        return [dict(src=pc['param']().data_ptr() + 4 * pc['offset'], fwd=e['wb'].data_ptr(), flip=e['wt'].data_ptr(),
                     strides=pc['strides'], dims=pc['dims'], dst=(cout, cin, pc['co_off'], pc['ci_off'])) for pc in e['pieces']]

    @staticmethod
    def _fill_table(rows):
        arr = (_lib.WprepEntry * len(rows))()
        block = 0
        for rec, r in zip(arr, rows):
            rec.src, rec.fwd, rec.flip = r['src'], r['fwd'], r['flip']
            rec.stride_co, rec.stride_ci, rec.stride_kh, rec.stride_kw = r['strides']
            rec.first_block = block
            rec.cout, rec.cin, rec.kh, rec.kw = r['dims']
            rec.dst_cout, rec.dst_cin, rec.co_off, rec.ci_off = r.get('dst', (0, 0, 0, 0))
            rec.fwd_f32 = r.get('fwd_f32', 0)
            block += (r['dims'][0] * r['dims'][1] * r['dims'][2] * r['dims'][3] + 255) // 256
        return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8), block

    def _build_table(self):
        rows = [r for e in self.order for r in self._rows(e)]
        host, self.total_blocks = self._fill_table(rows)
        self.table = host.to(self.device)
        self.rows = len(rows)

    def refresh(self):
        """Rewrite every shadow from its fp32 master: one launch."""
        if not self.order:
            return
        if not all(_shadow_alive(e) for e in self.order):
            self.order = [e for e in self.order if _shadow_alive(e)]
            self.entries = {id(e['ref']()): e for e in self.order if 'ref' in e}
            self.assembled = {e['key']: e for e in self.order if e.get('kind') == 'assembled'}
            if not self.order:
                return
            self._build_table()
        import contextlib
        # the launch goes to the device that owns the pointers in the table (CPU tensors: the kernels-on-CPU test build)
        guard = torch.cuda.device(self.device) if self.device.type == 'cuda' else contextlib.nullcontext()
        with guard:
            check(_lib.lib().stp3_conv2d_prep_weights(_ptr(self.table), self.rows, self.total_blocks, _stream()),
                  'stp3_conv2d_prep_weights')
        for e in self.order:
            if 'ref' in e:
                e['version'] = e['ref']()._version
            else:
                e['versions'] = tuple(pc['param']()._version for pc in e['pieces'])
            e.pop('phases', None)                  # sub-kernels cut from the flipped shadow (_strided_dgrad)

    # -- the gradients of the assembled weights: back to their parameters ---------------------------------------------------
    def scatter_rows(self, e):
        """Table rows that cut ``e['dw']`` into the parameters' gradient slices (``_stp3_grad_view``: same strides as the
        parameter, parallel.GradientBuckets)."""
        cout, cin = e['shape'][:2]
        return [dict(src=pc['param']()._stp3_grad_view.data_ptr() + 4 * pc['offset'], fwd=e['dw'].data_ptr(), flip=0,
                     strides=pc['strides'], dims=pc['dims'], dst=(cout, cin, pc['co_off'], pc['ci_off'])) for pc in e['pieces']]

    def scatter(self, ents):
        """One launch: the gradients of the assembled weights ``ents`` into their parameters' bucket slices.  The table of a
        set of weights is uploaded once and kept (the same set comes back every step; a captured step replays the launch)."""
        if not ents:
            return
        key = tuple(id(e) for e in ents) + tuple(pc['param']()._stp3_grad_view.data_ptr() for e in ents for pc in e['pieces'])
        tab = self.scatter_tables.get(key)
        if tab is None:
            rows = [r for e in ents for r in self.scatter_rows(e)]
            host, blocks = self._fill_table(rows)
            if len(self.scatter_tables) > 64:
                self.scatter_tables.clear()
            tab = self.scatter_tables[key] = (host.to(self.device), len(rows), blocks, list(ents))
        import contextlib
        guard = torch.cuda.device(self.device) if self.device.type == 'cuda' else contextlib.nullcontext()
        with guard:
            check(_lib.lib().stp3_conv2d_scatter_weight_grads(_ptr(tab[0]), tab[1], tab[2], _stream()),
                  'stp3_conv2d_scatter_weight_grads')


def _shadow_alive(e):
    if 'ref' in e:
        w = e['ref']()
        return w is not None and w.data_ptr() == e['ptr']
    return all(pc['param']() is not None and pc['param']().data_ptr() == pc['ptr'] for pc in e['pieces'])


def _pieces_signature(shape, pieces):
    return (tuple(shape),) + tuple((id(pc['param']()), pc['ptr'], pc['offset'], tuple(pc['strides']), tuple(pc['dims']),
                                    pc['co_off'], pc['ci_off']) for pc in pieces)


_SHADOW_TABLES = {}                                # device -> _WeightShadows


def _shadows(device):
    device = torch.device(device)
    tab = _SHADOW_TABLES.get(device)
    if tab is None:
        tab = _SHADOW_TABLES[device] = _WeightShadows(device)
    return tab


def weight_stamp(weight):
    """What identifies the VALUES a parameter's bf16 shadows were cut from: the parameter's version counter and the
    epoch of out-of-band updates (``invalidate_weight_cache``).  The autograd operators record it at forward time and
    compare at backward time: the shadows are rewritten in place behind autograd's back, so a weight update between a
    forward and its backward would otherwise silently differentiate against the NEW weights where torch raises
    'modified by an inplace operation'."""
    return (weight._version, _WEIGHT_EPOCH[0])


def check_weight_stamp(weight, stamp, what):
    if stamp is not None and weight_stamp(weight) != stamp:
        raise RuntimeError(f'{what}: the convolution weight was updated between this forward and its backward '
                           f'(version / epoch {stamp} -> {weight_stamp(weight)}); its bf16 shadow no longer holds the '
                           f'values the forward used')


def invalidate_weight_cache():
    """Call after updating parameters through storage the parameter's version counter does not see (the flat
    buffers of ``parallel.FlatAdam``); in-place updates of the parameters themselves are detected automatically."""
    _WEIGHT_EPOCH[0] += 1
    for tab in _SHADOW_TABLES.values():
        tab.refresh()


def _bf16_weights(weight, need_flipped=False):
    asm = getattr(weight, '_stp3_assembled', None)
    if asm is not None:                          # (``assembled_weight``: the stand-in tensor of a weight made of parameter views)
        return asm['wb'], asm['wt']
    if isinstance(weight, torch.nn.Parameter) and weight.requires_grad \
            and weight.dtype == torch.float32 and weight.is_cuda:
        tab = _shadows(weight.device)
        ent = tab.lookup(weight) or tab.register(weight)
        return ent['wb'], ent['wt']
    key = id(weight)
    ent = _WEIGHT_CACHE.get(key)
    ver = (weight._version, _WEIGHT_EPOCH[0])
    # id() values are recycled: an entry only counts when it still points at THIS live parameter (weak reference),
    # at the same storage, shape and strides
    if ent is not None and (ent[4]() is not weight or ent[0] != ver or ent[1] != weight.data_ptr()
                            or ent[5] != (tuple(weight.shape), tuple(weight.stride()))):
        ent = None
    if ent is None:
        wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ent = [ver, weight.data_ptr(), wb, None, None, None]
        if isinstance(weight, torch.nn.Parameter) and weight.requires_grad:
            # parameters only: temporaries (weight slices, padded copies) die with the call
            ent[4] = weakref.ref(weight)
            ent[5] = (tuple(weight.shape), tuple(weight.stride()))
            _WEIGHT_CACHE[key] = ent
            weakref.finalize(weight, _evict_weight, key, ent)
    if need_flipped and ent[3] is None:
        ent[3] = ent[2].flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
    return ent[2], ent[3]


# Weights the model ASSEMBLES from parameters -- zero-padded channel lanes, the taps of a causal 3-D kernel side by side, the
# heads of the decoder merged into one convolution, a projection split in two -- as shadows that stp3_conv2d_prep_weights writes
# piece by piece with all the others, and whose gradient one launch per backward pass cuts back into the parameters' bucket
# slices (``assembled_weight``).  Off: the call sites build the weight with torch (pad / cat / slice and their backward).
ASSEMBLED_WEIGHTS = True


def weight_piece(param, view, co_off=0, ci_off=0):
    """One piece of an assembled weight: ``view`` -- a 4-D (co, ci, kh, kw) view of the leaf parameter ``param`` (any strides:
    ``param.detach().unbind(2)[k]``, a channel split, the centre tap) -- placed at channel offsets (co_off, ci_off)."""
    assert view.dim() == 4 and view.dtype == torch.float32 and param.dtype == torch.float32
    off = (view.data_ptr() - param.data_ptr()) // 4
    return {'param': weakref.ref(param), 'ptr': param.data_ptr(), 'offset': off, 'strides': tuple(view.stride()),
            'dims': tuple(view.shape), 'co_off': int(co_off), 'ci_off': int(ci_off)}


def assembled_weight_supported(x, params):
    """The assembled route serves the bf16 kernels only (nothing holds the weight's float32 values): bf16 GPU activations (or
    autocast, which makes them so) and float32 leaf parameters on the same device."""
    return (ASSEMBLED_WEIGHTS and x.is_cuda and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled())
            and all(isinstance(p, torch.nn.Parameter) and p.is_leaf and p.dtype == torch.float32 and p.device == x.device
                    for p in params))


def assembled_weight(key, shape, pieces, direct=True):
    """The stand-in tensor of the (Cout, Cin, KH, KW) weight made of ``pieces`` (``weight_piece``), zero elsewhere: hand it to
    ``conv2d`` / ``ops_fused.conv_bn_act`` as the weight.  Its bf16 shadows exist from the first call on and are rewritten with
    every other weight's after an optimizer step; its gradient reaches the parameters of the pieces (through autograd: the
    stand-in is the output of ``_AssembledWeight``).  ``key``: what identifies the weight among the model's (a module and a
    name); the pieces may change between calls (another lane count): the entry is rebuilt then.
    ``direct=False``: a parameter of the pieces ALSO reaches the loss some other way: its gradient contributions must meet in
    autograd, so this one is handed over as a tensor.  ``direct='shared'``: that other way is ``weight_columns`` (the columns
    of a projection that multiply a pooled vector) and the two together cover the parameter: both write their part of the
    bucket slice, one of them hands autograd the alias."""
    params = []
    for pc in pieces:
        p = pc['param']()
        if not any(p is q for q in params):
            params.append(p)
    tab = _shadows(params[0].device)
    ent = tab.assembled.get(key)
    if ent is not None and (ent['signature'] != _pieces_signature(shape, pieces) or not _shadow_alive(ent)):
        tab.order = [e for e in tab.order if e is not ent]
        ent = None
    if ent is None:
        ent = tab.register_assembled(key, shape, pieces)
    elif ent['versions'] != tuple(pc['param']()._version for pc in ent['pieces']):
        tab.refresh()                              # a parameter was updated in place behind our back (a torch optimizer)
    ent['uses'] += 1
    token = _AssembledWeight.apply(ent, direct == 'shared', *params)
    token._stp3_assembled = ent
    token._stp3_assembled_direct = direct
    return token


def _direct_grad_ready(p, device):
    """A leaf parameter whose gradient an operator may write straight into its bucket slice (gather-mode buckets: see
    ``_conv2d_wgrad``)."""
    view = getattr(p, '_stp3_grad_view', None)
    return (view is not None and p.grad is None and p.requires_grad and view.dtype == torch.float32 and view.device == device
            and _same_memory_order(view, p))


def _entry_params(ent):
    return list({id(pc['param']()): pc['param']() for pc in ent['pieces']}.values())


def _direct_ok(ent, task, device, shared):
    """The ONE predicate every operator that writes a part of the gradients of an assembled weight's parameters evaluates
    (``_conv2d_wgrad`` for the weight itself, ``_WeightColumns`` for the columns that go their own way): all of them take the
    direct route or none does.  It only reads state that does not change between their backward calls."""
    if not (ASSEMBLED_WEIGHTS and DIRECT_BUCKET_GRADS) or task == -1 or ent['uses'] != 1:
        return False
    ranges = sorted(ent['columns'])
    if any(a[1] > b[0] for a, b in zip(ranges, ranges[1:])) or (ranges and not shared):
        return False                               # (the same columns used twice: their gradients must be added)
    for p in _entry_params(ent):
        if not _direct_grad_ready(p, device):
            return False
        if getattr(p, '_stp3_grad_claim', None) == task and not (shared and getattr(p, '_stp3_grad_shared', None) == id(ent)):
            return False                           # (claimed by an operator that is not part of this weight)
    return True


def _claim_params(ent, task, shared):
    """Mark the parameters as written directly in this pass; returns the ones nobody had claimed (their AccumulateGrad gets
    the alias of the bucket slice from the caller)."""
    first = set()
    for p in _entry_params(ent):
        if getattr(p, '_stp3_grad_claim', None) != task:
            first.add(id(p))
        p._stp3_grad_claim = task
        p._stp3_grad_shared = id(ent) if shared else None
    return first


def _assembled_claim(ent, dw, shared):
    """May the weight gradient of this backward pass go to the entry's buffer and from there to the bucket slices?  Yes for a
    weight applied once since ``zero_grad`` whose parameters all take a direct gradient (see ``_conv2d_wgrad``: gather-mode
    buckets, no gradient on the parameter yet) and are not claimed by another operator of this pass -- except, for a
    ``shared`` weight, by the operator that writes the parameter's OTHER columns (``weight_columns``)."""
    task = _graph_task_id()
    if ent['claim'] == task or tuple(dw.shape) != ent['shape'] or not _direct_ok(ent, task, dw.device, shared):
        return False
    ent['mine'] = _claim_params(ent, task, shared)
    ent['claim'] = task
    return True


class _WeightColumns(torch.autograd.Function):
    """Columns [c0, c1) of a 1x1 kernel parameter as a (Cout, c1 - c0, 1, 1) tensor (the values themselves: an alias) for the
    operators that read float32 weights (``small_linear``), when the parameter's OTHER columns are a piece of an assembled
    weight (``assembled_weight(..., direct='shared')``): backward stores the columns' gradient into the parameter's bucket
    slice and hands autograd an alias of the slice -- or nothing when the assembled weight already did."""

    @staticmethod
    def forward(ctx, param, ent, c0, c1):
        ctx.param, ctx.ent, ctx.cols = param, ent, (c0, c1)
        base = param.detach()
        return base.as_strided((base.shape[0], c1 - c0, 1, 1), (base.stride(0), base.stride(1), 1, 1),
                               base.storage_offset() + c0 * base.stride(1))

    @staticmethod
    def backward(ctx, g):
        p, ent, (c0, c1) = ctx.param, ctx.ent, ctx.cols
        task = _graph_task_id()
        if _direct_ok(ent, task, g.device, True):
            view = p._stp3_grad_view
            cols = view.as_strided((p.shape[0], c1 - c0, 1, 1), (view.stride(0), view.stride(1), 1, 1),
                                   view.storage_offset() + c0 * view.stride(1))
            if g.data_ptr() != cols.data_ptr():     # (``_SmallLinear`` writes its weight gradient there itself)
                cols.copy_(g)
            # (when the weight's own gradient comes later, ``_assembled_claim`` finds the parameter claimed for this weight and
            # hands out no second alias)
            first = _claim_params(ent, task, True)
            return (view.detach() if id(p) in first else None), None, None, None
        full = torch.zeros_like(p)
        full.as_strided((p.shape[0], c1 - c0, 1, 1), (full.stride(0), full.stride(1), 1, 1), c0 * full.stride(1)).copy_(g)
        return full, None, None, None


def weight_columns(token, param, c0, c1):
    """Columns [c0, c1) of ``param`` for a float32 operator; ``token``: the stand-in of the assembled weight that holds the
    parameter's other columns (``assembled_weight(..., direct='shared')``).  See ``_WeightColumns``."""
    ent = token._stp3_assembled
    ent['columns'].append((int(c0), int(c1)))
    cols = _WeightColumns.apply(param, ent, int(c0), int(c1))
    if len(_COLUMN_DESTS) > 256:
        _COLUMN_DESTS.clear()
    _COLUMN_DESTS[(cols.data_ptr(), cols.shape[0], cols.shape[1])] = (ent, weakref.ref(param), int(c0), int(c1))
    return cols


class _AssembledWeight(torch.autograd.Function):
    """parameters -> the stand-in tensor of the weight assembled from their views.  Forward launches nothing (the shadows are
    current).  Backward receives the gradient of the whole weight: when ``_conv2d_wgrad`` wrote it into the entry's buffer, the
    parameters get fresh aliases of their bucket slices (autograd keeps them as ``.grad`` without a launch) and the entry
    joins the pass's scatter launch (``flush_wgrad_reductions``; at once when the buckets are exchanged between ranks: the
    bucket hooks may send a bucket as soon as its last gradient has landed); otherwise each parameter's gradient is cut out
    with torch."""

    @staticmethod
    def forward(ctx, ent, shared, *params):
        ctx.ent = ent
        ctx.shared = shared
        ctx.params = params
        return ent['token'].detach()

    @staticmethod
    def backward(ctx, dw):
        ent, params = ctx.ent, ctx.params
        direct = ent['claim'] == _graph_task_id() and dw.data_ptr() == ent['dw'].data_ptr()
        grads = []
        if direct:
            tab = _shadows(ent['dw'].device)
            for p in params:
                covered = sum(pc['dims'][0] * pc['dims'][1] * pc['dims'][2] * pc['dims'][3] for pc in ent['pieces'] if pc['param']() is p)
                if covered < p.numel() and not ctx.shared and ent.get('zeroed') != (id(p), p._stp3_grad_view.data_ptr()):
                    # a dropped tap, an unused slice: its gradient is zero.  ONCE per bucket slice: nothing writes those
                    # elements afterwards but scalings (clipping, the mean over ranks, the optimizer's write-back of the clipped
                    # gradient) and the passes that take the other route, which store zeros there themselves
                    p._stp3_grad_view.zero_()
                    ent['zeroed'] = (id(p), p._stp3_grad_view.data_ptr())
                # (a shared parameter whose other columns were written first: their operator handed over the alias)
                grads.append(p._stp3_grad_view.detach() if id(p) in ent['mine'] else None)
            if _single_process():
                tab.pending_scatter.append(ent)
            else:
                tab.scatter([ent])
            return (None, None) + tuple(grads)
        for p, need in zip(params, ctx.needs_input_grad[2:]):
            if not need:
                grads.append(None)
                continue
            g = torch.zeros_like(p)
            for pc in ent['pieces']:
                if pc['param']() is p:
                    co, ci = pc['dims'][:2]
                    g.as_strided(pc['dims'], pc['strides'], pc['offset']).copy_(
                        dw[pc['co_off']:pc['co_off'] + co, pc['ci_off']:pc['ci_off'] + ci])
            grads.append(g)
        return (None, None) + tuple(grads)


def _evict_weight(key, ent):
    if _WEIGHT_CACHE.get(key) is ent:
        del _WEIGHT_CACHE[key]


def _phase_taps(k, pad, stride, phase):
    """Input rows hi = stride * i + phase of a strided convolution's data gradient: dx[hi] = sum over the taps kh with
    (phase + pad - kh) % stride == 0 of dy[i + (phase + pad - kh) / stride] * w[kh] -- a STRIDE-1 correlation of dy
    with every stride-th tap.  Returns (first index into the tap-FLIPPED kernel, number of taps, top padding) or None
    when no tap lands on this phase; the flipped-kernel taps first, first + stride, ... are in increasing dy offset."""
    khs = [kh for kh in range(k) if (phase + pad - kh) % stride == 0]
    if not khs:
        return None
    top = -((phase + pad - khs[-1]) // stride)            # largest kh: smallest dy offset
    return k - 1 - khs[-1], len(khs), top


def _strided_dgrad(dy, wt, x_shape, stride, pad, cache, out_dtype=torch.bfloat16):
    """dL/dx of a stride-s convolution WITHOUT zero-stuffing dy to the input resolution: one stride-1 convolution per
    input phase (s x s of them, each over dy at ITS resolution with the taps that land on the phase), results
    interleaved into dx.  The zero-stuffed form multiplies s^2 - 1 zeros out of s^2 (decoder stem 7x7 / 2: 370 us where
    the four phase convolutions need a quarter of the work).  ``wt`` (Cin,Cout,KH,KW): taps flipped, channels swapped.
    ``cache``: dict that keeps the phase sub-kernels of a parameter's shadow until the shadows are rewritten.
    Returns None when a phase would need negative padding (pad > what the taps reach): the caller zero-stuffs."""
    n, cin, h, w = x_shape
    kh, kw = wt.shape[2], wt.shape[3]
    plans = []
    for ph in range(stride):
        th = _phase_taps(kh, pad[0], stride, ph)
        for pw in range(stride):
            tw = _phase_taps(kw, pad[1], stride, pw)
            if th is not None and tw is not None and (th[2] < 0 or tw[2] < 0):
                return None
            plans.append((ph, pw, th, tw))
    dx = torch.empty((n, cin, h, w), dtype=out_dtype, device=dy.device, memory_format=torch.channels_last)
    for ph, pw, th, tw in plans:
        rows, cols = (h - ph + stride - 1) // stride, (w - pw + stride - 1) // stride
        if rows <= 0 or cols <= 0:
            continue
        if th is None or tw is None:
            dx[:, :, ph::stride, pw::stride] = 0
            continue
        key = (ph, pw)
        sub = None if cache is None else cache.get(key)
        if sub is None:
            sub = wt[:, :, th[0]::stride, tw[0]::stride].contiguous(memory_format=torch.channels_last)
            if cache is not None:
                cache[key] = sub
        part = _conv2d_launch(dy, sub, None, 1, (th[2], tw[2]), (1, 1), out_dtype, out_hw=(rows, cols))
        dx[:, :, ph::stride, pw::stride] = part
    return dx


def conv2d_data_grad(dy, wb, weight_ref, x_shape, stride, pad, dil, out_dtype=torch.bfloat16, add=None):
    """dL/dx of a dense convolution on the MFMA kernel: a stride-1 convolution of dy with the taps flipped and
    Cin / Cout swapped -- per input phase for a strided layer (``_strided_dgrad``), over the zero-stuffed dy when that
    does not apply.  dy (N,Cout,Ho,Wo) bf16 channels-last; wb (Cout,Cin,KH,KW) bf16; ``weight_ref``: the parameter wb
    shadows (its flipped shadow and phase sub-kernels are cached per optimizer step) or None."""
    cout, cin, kh, kw = wb.shape
    bpad = (dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1])
    phase_cache = None
    asm = getattr(weight_ref, '_stp3_assembled', None) if weight_ref is not None else None
    if asm is not None:
        wt = asm['wt']
        phase_cache = asm.setdefault('phases', {})
    elif weight_ref is not None and weight_ref.is_leaf and weight_ref.requires_grad:
        wt = _bf16_weights(weight_ref, need_flipped=True)[1]
        ent = _shadows(weight_ref.device).lookup(weight_ref) if isinstance(weight_ref, torch.nn.Parameter) else None
        if ent is not None and ent['wt'] is wt:
            phase_cache = ent.setdefault('phases', {})
    else:
        wt = wb if kh == kw == 1 else wb.flip(2, 3)              # a 1x1 kernel has nothing to flip
        wt = wt.transpose(0, 1).contiguous(memory_format=torch.channels_last)
    dx = None
    # per-phase sub-convolutions do 1 / stride^2 of the zero-stuffed form's multiplications in stride^2 launches plus
    # as many interleaving copies: measured on the MI355X (scripts/time_strided_dgrad.py) they win on the decoder's
    # 7x7 / 2 stem (361 -> 195 us, 193 GFLOP zero-stuffed) and lose 15-20 us on its 3x3 / 2 and 1x1 / 2 layers
    # (<= 18 GFLOP), so the choice goes by the zero-stuffed work
    if stride > 1 and tuple(dil) == (1, 1):
        stuffed_flops = 2.0 * x_shape[0] * x_shape[2] * x_shape[3] * cout * cin * kh * kw
        if stuffed_flops >= 5e10:
            dx = _strided_dgrad(dy, wt, x_shape, stride, pad, phase_cache, out_dtype)
    if dx is None:
        g = dy
        if stride > 1:
            n, _, h, w = x_shape
            ho, wo = dy.shape[2], dy.shape[3]
            # rows / columns the forward never reached (floor in the output-size formula) get zero gradient
            uh = h + 2 * pad[0] - dil[0] * (kh - 1)
            uw = w + 2 * pad[1] - dil[1] * (kw - 1)
            g = torch.empty((n, cout, uh, uw), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last).zero_()
            g[:, :, ::stride, ::stride][:, :, :ho, :wo] = dy
        return _conv2d_launch(g, wt, None, 1, bpad, dil, out_dtype, add=add)
    return dx if add is None else dx + add.to(dx.dtype)


def channel_sums(t):
    """Per-channel sums over (N, H, W) of a channels-last bf16 / float32 GPU tensor, float32 -- the bias gradient of a
    convolution -- through the statistics kernel of the BatchNorm family (stp3_bn_stats: a coalesced two-stage column
    reduction).  torch's own ``sum(dim=(0, 2, 3))`` of a channels-last tensor with a handful of channels takes a strided
    reduction path: 620 us for the (12, 16, 200, 200) gradient of the merged decoder heads, measured
    (profiles/r04c_step_trace.txt), against ~10 us here."""
    t, ld = _rows_view(t)
    n, c, h, w = t.shape
    per = 8 if t.dtype == torch.bfloat16 else 4
    if t.dtype not in (torch.bfloat16, torch.float32) or ld % per or c % per or t.data_ptr() % 16:
        return t.float().sum(dim=(0, 2, 3))
    dims = _lib.BnDims(n, h * w, c, ld, ld, ld, _lib.DTYPE_BF16 if t.dtype == torch.bfloat16 else _lib.DTYPE_F32, ACT_NONE,
                       RES_NONE, 0, 0, 0)
    ws, ws_bytes = _bn_workspace(n, c, t.device)
    sums = torch.empty(2 * c, dtype=torch.float32, device=t.device)
    check(_lib.lib().stp3_bn_stats(ctypes.byref(dims), t.data_ptr(), None, ws.data_ptr(), ws_bytes, sums.data_ptr(),
                                   _stream_handle()), 'stp3_bn_stats')
    return sums[:c]


class _Conv2dMfma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, out_dtype):
        _need_gpu(x, weight)
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        wb, _ = _bf16_weights(weight)
        fb = _f32(bias)
        y = _conv2d_launch(x, wb, fb, stride, pad, dil, out_dtype)
        ctx.weight_ref = note_weight_use(weight)
        ctx.weight_stamp = weight_stamp(weight)
        ctx.save_for_backward(x, wb)
        ctx.cfg = (stride, pad, dil, bias is not None, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        check_weight_stamp(ctx.weight_ref, ctx.weight_stamp, 'conv2d backward')
        stride, pad, dil, has_bias, wdtype, bdtype = ctx.cfg
        cout_true, cin, kh, kw = wb.shape
        dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        cpad = (-cout_true) % 8
        if cpad:
            # the 1 / 2 / 4-channel heads: dy and the weight get zero channels up to a multiple of 8, so that both
            # gradients run on the MFMA kernels too (16-byte channel pieces); the extra rows of dw are dropped
            dy = torch.nn.functional.pad(dy, (0, 0, 0, 0, 0, cpad)).contiguous(memory_format=torch.channels_last)
            wb = torch.nn.functional.pad(wb, (0, 0, 0, 0, 0, 0, 0, cpad)).contiguous(memory_format=torch.channels_last)
        db = None
        if has_bias and ctx.needs_input_grad[2]:
            db = channel_sums(dy)[:cout_true].to(bdtype)              # (of the bf16 gradient, as before; padding lanes are zero)
        cout = cout_true + cpad
        dx = dw = None
        bpad = (dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1])
        need_dx = ctx.needs_input_grad[0]
        hip_dx = need_dx and cout % 8 == 0 and bpad[0] >= 0 and bpad[1] >= 0
        if hip_dx:
            dx = conv2d_data_grad(dy, wb, None if cpad else ctx.weight_ref, x.shape, stride, pad, dil)
        need_dw = ctx.needs_input_grad[1]
        hip_dw = need_dw and cin % 8 == 0
        if hip_dw:
            dw = _conv2d_wgrad(dy, x, (cout, cin, kh, kw), stride, pad, dil, leaf=None if cpad else ctx.weight_ref)[:cout_true].to(wdtype)
        mask = [need_dx and not hip_dx, need_dw and not hip_dw, False]
        if any(mask):
            # not reached by the model (every layer satisfies the kernels' constraints); kept so that an odd
            # convolution still differentiates
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            gx, gw, _ = torch.ops.aten.convolution_backward(dy, xd, wb, None, [stride, stride], list(pad), list(dil), False,
                                                            [0, 0], 1, mask)
            if mask[0]:
                dx = gx
            if mask[1]:
                dw = gw[:cout_true].to(wdtype)
        return dx, dw, db, None, None, None, None


_CONV_APPLY = _fast_apply(_Conv2dMfma)


# ----------------------------------------------------------------------------------------------
# float32 convolutions on the SAME matrix-core kernels: three-term bf16 split
# ----------------------------------------------------------------------------------------------
def _split3(t):
    """float32 t = hi + mid + lo exactly up to 2^-24 |t|, every term a bf16 number (8 significant bits each: the
    subtractions are exact in float32)."""
    t = t.float()
    hi = t.to(torch.bfloat16)
    r = t - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


# (a, b) term pairs of a product of two split operands, smallest first; the dropped ones (mid*lo, lo*mid, lo*lo) are
# below 2^-24 of |a| |b|, i.e. below float32's own rounding of the product
_SPLIT_PAIRS = ((1, 1), (0, 2), (2, 0), (0, 1), (1, 0), (0, 0))


class _Conv2dSplit3(torch.autograd.Function):
    """A FLOAT32 dense convolution -- forward, data gradient, weight gradient -- on the bf16 MFMA kernels of
    stp3_conv.hip (``conv2d_igemm_kernel``, ``conv2d_wgrad_kernel``), float32-accurate: both operands are split into
    three bf16 terms and the six significant term products are accumulated in float32 (the kernels' own accumulators,
    float32 outputs).  Six launches per pass instead of one: this is not a fast path, it is how float32 tensors outside
    autocast -- the float32 legs of the parity tests, which pin the step against the reference's float32 / float64
    runs to 1e-5..1e-3 -- exercise the very kernels the bf16 step runs on, at the step's real shapes, instead of a
    vendor convolution."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil):
        _need_gpu(x, weight)
        xs = _split3(x.detach())
        ws = [t.contiguous(memory_format=torch.channels_last) for t in _split3(weight.detach())]
        fb = _f32(bias)
        y = None
        for k, (i, j) in enumerate(_SPLIT_PAIRS):
            last = k == len(_SPLIT_PAIRS) - 1
            part = _conv2d_launch(xs[i], ws[j], fb if last else None, stride, pad, dil, torch.float32)
            y = part if y is None else y.add_(part)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, dil, bias is not None, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, dil, has_bias, bdtype = ctx.cfg
        cout_true, cin, kh, kw = weight.shape
        dy = dy.float()
        db = dy.sum(dim=(0, 2, 3)).to(bdtype) if has_bias and ctx.needs_input_grad[2] else None
        w32 = weight.detach().float()
        cpad = (-cout_true) % 8
        if cpad:                                            # the 1 / 2 / 4-channel heads (see _Conv2dMfma.backward)
            dy = torch.nn.functional.pad(dy, (0, 0, 0, 0, 0, cpad))
            w32 = torch.nn.functional.pad(w32, (0, 0, 0, 0, 0, 0, 0, cpad))
        cout = cout_true + cpad
        dys = [t.contiguous(memory_format=torch.channels_last) for t in _split3(dy)]
        dx = dw = None
        bpad = (dil[0] * (kh - 1) - pad[0], dil[1] * (kw - 1) - pad[1])
        if ctx.needs_input_grad[0]:
            if bpad[0] < 0 or bpad[1] < 0:
                raise _lib.Stp3HipError('conv2d (float32 route): padding larger than the kernel reach')
            ws = [t.contiguous(memory_format=torch.channels_last) for t in _split3(w32)]
            for i, j in _SPLIT_PAIRS:
                part = conv2d_data_grad(dys[i], ws[j], None, x.shape, stride, pad, dil, out_dtype=torch.float32)
                dx = part if dx is None else dx.add_(part)
        if ctx.needs_input_grad[1]:
            xs = _split3(x.detach())
            for i, j in _SPLIT_PAIRS:
                part = _conv2d_wgrad(dys[i], xs[j], (cout, cin, kh, kw), stride, pad, dil)
                dw = part if dw is None else dw.add_(part)
            dw = dw[:cout_true].to(weight.dtype)
        return dx, dw, db, None, None, None


def conv2d_f32(x, weight, bias=None, stride=1, padding=0, dilation=1):
    """float32 in, float32 out, float32-accurate, on the MFMA kernels (``_Conv2dSplit3``)."""
    s = _pair(stride)
    return _Conv2dSplit3.apply(x, weight, bias, s[0], _pair(padding), _pair(dilation))


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, out_dtype=torch.bfloat16):
    """Dense conv through the MFMA implicit-GEMM kernel (bf16 operands, float32 accumulation)."""
    s = _pair(stride)
    return _CONV_APPLY(x, weight, bias, s[0], _pair(padding), _pair(dilation), out_dtype)
