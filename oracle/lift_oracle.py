"""TEST INFRASTRUCTURE -- CPU restatement of the reference's LSS lift + voxel-pool path.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import it.
It restates, on the CPU, what ``stp3/models/stp3.py`` and
``stp3/utils/geometry.py`` of the reference compute on the path

    create_frustum -> get_geometry -> softmax(depth) (x) feat -> ego alignment ->
    voxel index -> per-voxel sum (VoxelsSumming) -> discounted accumulation over T

Every function cites the reference lines it follows.  The integer part (voxel
ids) is written with numpy float32 arithmetic in an explicit operation order --
one IEEE rounding per operation, no FMA -- because that is what the reference's
torch-CPU ops evaluate to (verified bitwise against the reference itself by
``oracle/make_golden.py`` in the build container; results in
``tests/golden/MANIFEST.json``).

Parity pin: the reference ships no tests or golden vectors,
so this oracle is pinned against *outputs of the reference itself run here*:
``oracle/make_golden.py`` imports the unmodified reference (through
``oracle/ref_stubs.py``), asserts agreement with the functions below and writes
the fixtures the tests replay on the GPU box.
"""
import numpy as np
import torch

F32 = np.float32


# --------------------------------------------------------------------------------------
# BEV grid / frustum (host-side constants)
# --------------------------------------------------------------------------------------
def bev_parameters(x_bound, y_bound, z_bound):
    """reference stp3/utils/geometry.py:40-59 (calculate_birds_eye_view_parameters)."""
    rows = [x_bound, y_bound, z_bound]
    res = torch.tensor([r[2] for r in rows])
    start = torch.tensor([r[0] + r[2] / 2.0 for r in rows])
    dim = torch.tensor([(r[1] - r[0]) / r[2] for r in rows], dtype=torch.long)
    return res, start, dim


def create_frustum(final_dim, downsample, d_bound):
    """reference stp3/models/stp3.py:111-130.  Returns (D, fH, fW, 3) float32 of (x_px, y_px, depth)."""
    h, w = final_dim
    fh, fw = h // downsample, w // downsample
    depth = torch.arange(*d_bound, dtype=torch.float)
    n_d = depth.shape[0]
    depth = depth.view(-1, 1, 1).expand(-1, fh, fw)
    xs = torch.linspace(0, w - 1, fw, dtype=torch.float).view(1, 1, fw).expand(n_d, fh, fw)
    ys = torch.linspace(0, h - 1, fh, dtype=torch.float).view(1, fh, 1).expand(n_d, fh, fw)
    return torch.stack((xs, ys, depth), -1).contiguous()


def camera_matrices(intrinsics, extrinsics):
    """reference stp3.py:189,196: combined = R . K^-1 (torch CPU ops), translation.

    intrinsics (..., 3, 3), extrinsics (..., 4, 4) -> M (..., 3, 3), trans (..., 3) float32.
    """
    rotation, translation = extrinsics[..., :3, :3], extrinsics[..., :3, 3]
    return rotation.matmul(torch.inverse(intrinsics)).contiguous(), translation.contiguous()


def ego_matrices(future_egomotion):
    """reference geometry.py:124-172 (euler2mat / pose_vec2mat): R = X(rx) . Y(ry) . Z(rz).

    future_egomotion (..., 6) = (tx,ty,tz,rx,ry,rz) -> R (..., 3, 3), t (..., 3) float32.
    """
    shape = future_egomotion.shape[:-1]
    ang = future_egomotion[..., 3:].contiguous().view(-1, 3)
    x, y, z = ang[:, 0], ang[:, 1], ang[:, 2]
    zeros, ones = torch.zeros_like(z), torch.ones_like(z)
    cz, sz = torch.cos(z), torch.sin(z)
    zmat = torch.stack([cz, -sz, zeros, sz, cz, zeros, zeros, zeros, ones], dim=1).view(-1, 3, 3)
    cy, sy = torch.cos(y), torch.sin(y)
    ymat = torch.stack([cy, zeros, sy, zeros, ones, zeros, -sy, zeros, cy], dim=1).view(-1, 3, 3)
    cx, sx = torch.cos(x), torch.sin(x)
    xmat = torch.stack([ones, zeros, zeros, zeros, cx, -sx, zeros, sx, cx], dim=1).view(-1, 3, 3)
    rot = xmat.bmm(ymat).bmm(zmat).view(*shape, 3, 3)
    return rot.contiguous(), future_egomotion[..., :3].contiguous()


# --------------------------------------------------------------------------------------
# Geometry -> voxel ids (integer result; bit-exact contract)
# --------------------------------------------------------------------------------------
def _affine3(mat, vec, x, y, z):
    """q_i = ((m_i0*x + m_i1*y) + m_i2*z) + v_i in float32, one rounding per op.

    This is what torch-CPU's small-matrix bmm followed by an in-place add evaluates to
    (reference stp3.py:197-198 and :275-276).  mat (..., 3, 3) broadcast against x/y/z.
    """
    out = []
    for i in range(3):
        acc = mat[..., i, 0] * x
        acc = acc + mat[..., i, 1] * y
        acc = acc + mat[..., i, 2] * z
        out.append(acc + vec[..., i])
    return out


def geometry_points(frustum, cam_m, cam_t):
    """reference stp3.py:186-201 (get_geometry).

    frustum (D,fH,fW,3); cam_m (BT,N,3,3); cam_t (BT,N,3)  ->  (BT,N,D,fH,fW,3) float32 ego-frame xyz.
    """
    fr = np.asarray(frustum, dtype=F32)
    m = np.asarray(cam_m, dtype=F32)[:, :, None, None, None]
    t = np.asarray(cam_t, dtype=F32)[:, :, None, None, None]
    d = fr[..., 2]
    px = (fr[..., 0] * d)[None, None]          # stp3.py:195
    py = (fr[..., 1] * d)[None, None]
    pz = d[None, None]
    q = _affine3(m, t, px, py, pz)
    return np.stack(q, axis=-1)


def ego_align(points, ego_r, ego_t):
    """reference stp3.py:263-277: frame k is mapped by ego[k], ego[k+1], ..., ego[S-2] (in that
    order, each a separate float32 affine) into the present frame; the present frame is untouched.

    points (B,S,N,D,fH,fW,3) float32 (a copy is returned); ego_r (B,S,3,3); ego_t (B,S,3).
    """
    pts = np.array(points, dtype=F32, copy=True)
    r = np.asarray(ego_r, dtype=F32)
    tr = np.asarray(ego_t, dtype=F32)
    b_, s_ = pts.shape[:2]
    for b in range(b_):
        for t in range(s_ - 1):
            sub = pts[b, :t + 1]
            m = r[b, t][None, None, None, None, None]
            v = tr[b, t][None, None, None, None, None]
            q = _affine3(m, v, sub[..., 0], sub[..., 1], sub[..., 2])
            pts[b, :t + 1] = np.stack(q, axis=-1)
    return pts


def bev_offset(bev_start, bev_res):
    """stp3.py:288: (bev_start_position - bev_resolution / 2.0), evaluated in float32."""
    return (bev_start.float() - bev_res.float() / 2.0).numpy().astype(F32)


def voxel_index(points, offset, res, dim):
    """reference stp3.py:287-289 + voxel_to_pixel mask/rank :239-255.

    ((p - offset) / res) truncated toward zero (``.long()``), in-range mask, rank =
    ix*(Y*Z) + iy*Z + iz.  Returns int32 (same leading shape as points[..., 0]) with -1 for
    points outside the grid.
    """
    p = np.asarray(points, dtype=F32)
    off = np.asarray(offset, dtype=F32)
    rs = np.asarray(res, dtype=F32)
    dm = [int(v) for v in dim]
    with np.errstate(invalid='ignore', over='ignore'):
        q = (p - off) / rs
        finite = np.isfinite(q).all(axis=-1)
        qi = np.where(np.isfinite(q), np.trunc(q), -1.0)
        keep = finite
        for a in range(3):
            keep = keep & (qi[..., a] >= 0) & (qi[..., a] < dm[a])
        qi = np.where(keep[..., None], qi, 0.0).astype(np.int64)
    rank = qi[..., 0] * (dm[1] * dm[2]) + qi[..., 1] * dm[2] + qi[..., 2]
    return np.where(keep, rank, -1).astype(np.int32)


def lift_voxel_ids(frustum, intrinsics, extrinsics, future_egomotion, x_bound, y_bound, z_bound):
    """Whole index path for (B,S,N,...) inputs -> int32 (B,S,N,D,fH,fW) voxel ids (-1 = dropped)."""
    b, s = intrinsics.shape[:2]
    res, start, dim = bev_parameters(x_bound, y_bound, z_bound)
    cam_m, cam_t = camera_matrices(intrinsics.reshape(b * s, *intrinsics.shape[2:]),
                                   extrinsics.reshape(b * s, *extrinsics.shape[2:]))
    ego_r, ego_t = ego_matrices(future_egomotion)
    pts = geometry_points(frustum.numpy(), cam_m.numpy(), cam_t.numpy())
    pts = pts.reshape(b, s, *pts.shape[1:])
    pts = ego_align(pts, ego_r.numpy(), ego_t.numpy())
    return voxel_index(pts, bev_offset(start, res), res.numpy(), dim.tolist())


# --------------------------------------------------------------------------------------
# Pooling (floating point)
# --------------------------------------------------------------------------------------
def depth_softmax(depth_logits):
    """reference stp3.py:215: softmax over the depth-bin axis.  (..., N, D, fH, fW)."""
    return torch.softmax(depth_logits, dim=-3)


def pool_reference_style(feat, depth_logits, vox, bev_dim, discount):
    """Literal CPU restatement of the reference's pooling arithmetic, including its lossy
    prefix-sum trick.  reference stp3.py:215-221 (outer product), :247-260 (mask, argsort),
    geometry.py:302-318 (VoxelsSumming: cumsum, keep last of run, adjacent difference),
    stp3.py:292-299 (scatter into zeros, ``bev*discount + tmp``, permute).

    feat (B,S,N,C,fH,fW) f32; depth_logits (B,S,N,D,fH,fW) f32; vox (B,S,N,D,fH,fW) int32 ids.
    Returns (B,S,C,X,Y) float32.
    """
    b_, s_, n_, c_, fh, fw = feat.shape
    d_ = depth_logits.shape[3]
    x_, y_ = int(bev_dim[0]), int(bev_dim[1])
    out = torch.zeros(b_, s_, c_, x_, y_, dtype=torch.float)
    vox_t = torch.as_tensor(np.asarray(vox)).long()
    # stp3.py:207-216: the softmax runs on the packed (B*S*N, D, fH, fW) tensor
    prob_all = depth_logits.reshape(b_ * s_ * n_, d_, fh, fw).softmax(dim=1).view(b_, s_, n_, d_, fh, fw)
    for b in range(b_):
        bev = torch.zeros(x_ * y_, c_)
        for t in range(s_):
            prob = prob_all[b, t]                                           # (N,D,fH,fW)
            lifted = prob.unsqueeze(1) * feat[b, t].unsqueeze(2)            # (N,C,D,fH,fW)
            x_b = lifted.permute(0, 2, 3, 4, 1).reshape(-1, c_)             # (N*D*fH*fW, C)
            ranks = vox_t[b, t].reshape(-1)
            mask = ranks >= 0
            x_b, ranks = x_b[mask], ranks[mask]
            order = ranks.argsort()
            x_b, ranks = x_b[order], ranks[order]
            tmp = torch.zeros(x_ * y_, c_)
            if x_b.shape[0] > 0:
                cs = x_b.cumsum(0)
                last = torch.ones(cs.shape[0], dtype=torch.bool)
                last[:-1] = ranks[1:] != ranks[:-1]
                cs, kept = cs[last], ranks[last]
                sums = torch.cat((cs[:1], cs[1:] - cs[:-1]))
                tmp[kept] = sums
            bev = bev * discount + tmp
            out[b, t] = bev.t().reshape(c_, x_, y_)
    return out


def pool_exact(feat, depth_logits, vox, bev_dim, discount, dtype=torch.float64):
    """Same quantity as ``pool_reference_style`` with an exact (float64) per-voxel sum:
    out[b,t] = sum_{k<=t} discount^(t-k) * Pool_k,  Pool_k[c,v] = sum_{p: vox(p)=v} prob[p]*feat[c,pix(p)].
    Used for tight-tolerance checks of the HIP kernel (hard part 3).
    """
    b_, s_, n_, c_, fh, fw = feat.shape
    x_, y_ = int(bev_dim[0]), int(bev_dim[1])
    out = torch.zeros(b_, s_, c_, x_, y_, dtype=dtype)
    vox_t = torch.as_tensor(np.asarray(vox)).long()
    for b in range(b_):
        bev = torch.zeros(x_ * y_, c_, dtype=dtype)
        for t in range(s_):
            prob = depth_logits[b, t].to(dtype).softmax(dim=1)
            lifted = prob.unsqueeze(1) * feat[b, t].to(dtype).unsqueeze(2)
            x_b = lifted.permute(0, 2, 3, 4, 1).reshape(-1, c_)
            ranks = vox_t[b, t].reshape(-1)
            mask = ranks >= 0
            tmp = torch.zeros(x_ * y_, c_, dtype=dtype)
            tmp.index_add_(0, ranks[mask], x_b[mask])
            bev = bev * discount + tmp
            out[b, t] = bev.t().reshape(c_, x_, y_)
    return out


def pool_backward_exact(grad_out, feat, depth_logits, vox, discount, dtype=torch.float64):
    """Closed-form gradient of ``pool_exact`` w.r.t. feat and depth logits:

    G_t = sum_{t'>=t} discount^(t'-t) dL/dout[b,t'];  dprob[n,d,h,w] = sum_c feat[n,c,h,w] G_t[c,v(p)];
    dfeat[n,c,h,w] = sum_d prob[n,d,h,w] G_t[c,v(p)];  dlogit = prob * (dprob - sum_d prob*dprob).
    Equals autograd through reference stp3.py:215-301 / geometry.py:320-330.
    """
    b_, s_, n_, c_, fh, fw = feat.shape
    d_ = depth_logits.shape[3]
    vox_t = torch.as_tensor(np.asarray(vox)).long()
    gfeat = torch.zeros(feat.shape, dtype=dtype)
    glogit = torch.zeros(depth_logits.shape, dtype=dtype)
    go = grad_out.to(dtype).reshape(b_, s_, c_, -1)
    for b in range(b_):
        g_acc = torch.zeros(c_, go.shape[-1], dtype=dtype)
        for t in reversed(range(s_)):
            g_acc = g_acc * discount + go[b, t]
            ids = vox_t[b, t]                                              # (N,D,fH,fW)
            valid = (ids >= 0)
            gathered = g_acc.t()[ids.clamp(min=0)]                         # (N,D,fH,fW,C)
            gathered = gathered * valid.unsqueeze(-1)
            prob = depth_logits[b, t].to(dtype).softmax(dim=1)             # (N,D,fH,fW)
            f = feat[b, t].to(dtype)                                       # (N,C,fH,fW)
            dprob = torch.einsum('ndhwc,nchw->ndhw', gathered, f)
            gfeat[b, t] = torch.einsum('ndhwc,ndhw->nchw', gathered, prob)
            glogit[b, t] = prob * (dprob - (prob * dprob).sum(dim=1, keepdim=True))
    return gfeat, glogit


def voxels_summing(x, geometry, ranks):
    """Operator-level restatement of the reference's ``VoxelsSumming.forward``
    (stp3/utils/geometry.py:302-318) with exact per-voxel sums: rows of ``x`` (M,C), already sorted
    by rank, are summed (float64) over every run of equal consecutive ``ranks``; ``geometry`` keeps
    the LAST row of each run (the rows the reference's mask ``ranks[1:] != ranks[:-1]`` selects).
    -> (x_sum (V',C) float64, geometry_kept (V',...), seg_off (V'+1,) int64)."""
    x = np.asarray(x, dtype=np.float64)
    ranks = np.asarray(ranks)
    m = x.shape[0]
    kept = np.ones(m, dtype=bool)
    if m > 1:
        kept[:-1] = ranks[1:] != ranks[:-1]                              # geometry.py:308-309
    ends = np.nonzero(kept)[0]
    seg_off = np.concatenate([[0], ends + 1]).astype(np.int64)
    out = np.zeros((len(ends), x.shape[1]), dtype=np.float64)
    for s in range(len(ends)):
        out[s] = x[seg_off[s]:seg_off[s + 1]].sum(axis=0)
    return out, np.asarray(geometry)[kept], seg_off


def voxels_summing_backward(grad_x, seg_off):
    """``VoxelsSumming.backward`` (geometry.py:320-330): every input row receives the gradient row
    of the voxel it was summed into."""
    grad_x = np.asarray(grad_x, dtype=np.float64)
    seg_off = np.asarray(seg_off)
    return np.repeat(grad_x, np.diff(seg_off), axis=0)


def egomotion_planes(future_egomotion, receptive_field, bev_hw):
    """reference stp3.py:145-152: six broadcast ego-motion planes, shifted by one frame
    (frame 0 gets zeros, frame t gets future_egomotion[t-1]).  -> (B,S,6,X,Y)."""
    b, s, c = future_egomotion.shape
    h, w = bev_hw
    sp = future_egomotion.view(b, s, c, 1, 1).expand(b, s, c, h, w)
    return torch.cat([torch.zeros_like(sp[:, :1]), sp[:, :receptive_field - 1]], dim=1)
