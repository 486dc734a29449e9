"""TEST INFRASTRUCTURE (container only) -- make the reference importable.

/root/reference is pure Python but depends on packages that are not installed
here (pyquaternion, nuscenes, timm, efficientnet_pytorch, torchvision, skimage,
fvcore, pytorch_lightning).  ``install()`` registers minimal stand-ins in
``sys.modules`` so that the reference's *own* first-party
arithmetic can be executed unmodified and used to pin the restatement in
``oracle/lift_oracle.py`` and to generate ``tests/golden`` fixtures
(``oracle/make_golden.py``).

Nothing under ``st-p3_amd/`` imports this file.  /root/reference does not exist on
the GPU box: there the only importer is ``bench.py``'s ``cpu_baseline`` leg, which
finds the archive ``__graft_entry__.build()`` left under ``oracle/_ref``
(``oracle/snapshot_reference.py``; git-ignored).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('STP3_REFERENCE_ROOT', '/root/reference')     # (tests point it elsewhere to exercise the archive)
SNAPSHOT_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')     # oracle/snapshot_reference.py


def reference_root():
    """Where the reference's ``stp3`` package can be imported from: the mounted tree in the build container, else the
    verified archive ``__graft_entry__.build()`` left under oracle/_ref (what the GPU box has), else None."""
    if os.path.isdir(os.path.join(REFERENCE_ROOT, 'stp3')):
        return REFERENCE_ROOT
    from oracle import snapshot_reference
    if snapshot_reference.verify(SNAPSHOT_ROOT):
        return snapshot_reference.archive_path(SNAPSHOT_ROOT)          # a zip archive: imported through zipimport
    return None


def reference_available():
    return reference_root() is not None


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def polygon(r, c, shape=None):
    """Restatement of ``skimage.draw.polygon`` (scikit-image is a dependency of the reference that is NOT installed
    here: stp3/cost.py:7, stp3/metrics.py:12; the reference's environment.yml pins scikit-image=0.18.1).  Published algorithm (skimage/draw/_draw.pyx ``_polygon`` + skimage/_shared/geometry.pxd
    ``point_in_polygon``): every integer (row, column) of the vertices' bounding box -- rows max(0, floor-int of
    min r) .. ceil(max r), the same for columns -- is tested with the even-odd crossing rule; points are emitted
    row by row, columns ascending.  This is the rule of the pinned 0.18.1; releases >= 0.19 additionally count points
    lying EXACTLY on an edge or vertex as inside -- the two agree whenever no integer point lies on the boundary, which
    holds for every footprint the default configuration builds (non-integer box corners).
    Anchor in the reference itself: stp3/metrics.py:313 documents 32 footprint cells for the default ego box."""
    import numpy as np
    r = np.asarray(r, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    minr, maxr = int(max(0, r.min())), int(np.ceil(r.max()))
    minc, maxc = int(max(0, c.min())), int(np.ceil(c.max()))
    if shape is not None:
        maxr, maxc = min(shape[0] - 1, maxr), min(shape[1] - 1, maxc)
    rr, cc = [], []
    n = len(r)
    for y in range(minr, maxr + 1):
        for x in range(minc, maxc + 1):
            inside, j = False, n - 1
            for i in range(n):
                if ((r[i] <= y < r[j]) or (r[j] <= y < r[i])) and x < (c[j] - c[i]) * (y - r[i]) / (r[j] - r[i]) + c[i]:
                    inside = not inside
                j = i
            if inside:
                rr.append(y)
                cc.append(x)
    return np.array(rr, dtype=np.intp), np.array(cc, dtype=np.intp)


def install(efficientnet_cls=None, resnet18_fn=None):
    """Register stubs and put the reference on sys.path.  Idempotent."""
    import numpy as np
    import torch.nn as nn

    root = reference_root()
    if root is None:
        raise RuntimeError('reference tree not present (neither /root/reference nor a verified oracle/_ref snapshot)')
    if not hasattr(np, 'int'):
        np.int = int  # encoder.py:28,84 uses the removed alias

    class _Dummy:  # placeholder for names that are imported (and sometimes constructed at import
        # time, tools.py:161-172) but never *used* on this path
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise RuntimeError('stubbed third-party symbol was called')

    _mod('pyquaternion', Quaternion=_Dummy)
    _mod('nuscenes')
    _mod('nuscenes.utils')
    _mod('nuscenes.utils.geometry_utils', transform_matrix=_Dummy)
    _mod('nuscenes.utils.data_classes', LidarPointCloud=_Dummy, Box=_Dummy)
    _mod('nuscenes.map_expansion')
    _mod('nuscenes.map_expansion.map_api', NuScenesMap=_Dummy)
    _mod('timm')
    _mod('timm.models')
    _mod('timm.models.layers', DropPath=nn.Identity)
    _mod('skimage')
    _mod('skimage.draw', polygon=polygon)

    class _Normalize:  # network.py:33 subclasses it
        def __init__(self, mean=None, std=None):
            self.mean, self.std = mean, std

    tv = _mod('torchvision')
    tv.transforms = _mod('torchvision.transforms', Normalize=_Normalize, Compose=_Dummy, ToTensor=_Dummy,
                         ToPILImage=_Dummy)
    tv.models = _mod('torchvision.models')
    tv.models.resnet = _mod('torchvision.models.resnet', resnet18=resnet18_fn or _Dummy)
    _mod('efficientnet_pytorch', EfficientNet=efficientnet_cls or _Dummy)

    if root not in sys.path:
        sys.path.insert(0, root)


def make_reference_lifter(final_dim=(224, 480), x_bound=(-50.0, 50.0, 0.5), y_bound=(-50.0, 50.0, 0.5),
                          z_bound=(-10.0, 10.0, 20.0), d_bound=(2.0, 50.0, 1.0), downsample=8, out_channels=64,
                          discount=0.5):
    """An ``STP3`` instance with only the lifting attributes populated (no encoder).

    Follows SURVEY.md Appendix C: built with ``__new__`` so that ``create_frustum``,
    ``get_geometry`` and ``projection_to_birds_eye_view`` (reference
    stp3/models/stp3.py:111-130, 186-201, 226-301) can be executed directly.
    """
    install()
    import torch.nn as nn
    from types import SimpleNamespace as NS
    from stp3.models.stp3 import STP3
    from stp3.utils.geometry import calculate_birds_eye_view_parameters

    m = STP3.__new__(STP3)
    nn.Module.__init__(m)
    m.cfg = NS(
        IMAGE=NS(FINAL_DIM=tuple(final_dim)),
        LIFT=NS(X_BOUND=list(x_bound), Y_BOUND=list(y_bound), Z_BOUND=list(z_bound), D_BOUND=list(d_bound),
                DISCOUNT=discount),
        MODEL=NS(ENCODER=NS(DOWNSAMPLE=downsample, OUT_CHANNELS=out_channels, USE_DEPTH_DISTRIBUTION=True),
                 TEMPORAL_MODEL=NS(INPUT_EGOPOSE=True)),
        PLANNING=NS(ENABLED=False),
    )
    res, start, dim = calculate_birds_eye_view_parameters(m.cfg.LIFT.X_BOUND, m.cfg.LIFT.Y_BOUND, m.cfg.LIFT.Z_BOUND)
    m.bev_resolution = nn.Parameter(res, requires_grad=False)
    m.bev_start_position = nn.Parameter(start, requires_grad=False)
    m.bev_dimension = nn.Parameter(dim, requires_grad=False)
    m.encoder_downsample = downsample
    m.encoder_out_channels = out_channels
    m.frustum = m.create_frustum()
    m.depth_channels = m.frustum.shape[0]
    m.discount = discount
    return m
