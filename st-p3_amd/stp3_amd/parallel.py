"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference trains through PyTorch-Lightning's ``accelerator='ddp'`` with
``sync_batchnorm=True`` and ``gradient_clip_val`` (train.py:43-56); the samples of a batch are
independent (the voxel pool loops ``for b in range(batch)``, stp3.py:265), so the path shards by
batch with no data-path collective -- only the gradient all-reduce and the BatchNorm statistics.

``GradientBuckets`` is a small purpose-built reducer rather than a generic DDP wrapper:
  * parameters are bucketed in *reverse registration order* (decoder -> temporal model -> encoder
    heads -> trunk), which is the order backward produces their gradients;
  * every ``.grad`` is a view into its bucket's flat buffer, so there is no gather/scatter copy;
  * a bucket's all-reduce (RCCL ``ncclAllReduce`` on the communication stream) is launched from the
    post-accumulate hook of the last gradient that lands in it, i.e. it overlaps the rest of the
    backward pass;
  * bucket size defaults to 8 MiB: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring
    all-reduce of 33 MB of fp32 gradients is per-link bound at ~0.4 ms, so a handful of large
    buckets keeps the link busy without serialising the tail behind one big transfer.
Works with the ``gloo`` backend on CPU tensors too (that is how tests/ cover world_size 2).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


# clipping + Adam through stp3_optim_clip_adam (csrc/stp3_optim.hip) on GPU buckets, see FlatAdam.clip_and_step
FUSED_ADAM = True


def init_distributed(backend=None):
    """Initialise ``torch.distributed`` from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'   # "nccl" is RCCL on ROCm
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def convert_sync_batchnorm(module, enabled=True):
    """Cross-replica BatchNorm on / off for every BatchNorm layer of ``module`` (train.py:47 ``sync_batchnorm=True``:
    the batch statistics are those of the global batch, so N GPUs x B/N samples equal 1 GPU x B samples).

    Unlike ``nn.SyncBatchNorm.convert_sync_batchnorm`` no module is replaced: the statistics exchange is part of the
    product's own BatchNorm operator (``layers.fused.bn_act`` / ``ops_fused.conv_bn_act``: one all-reduce of the
    per-channel sums per layer and direction), which every BatchNorm module already runs through and which is ON by
    default whenever ``torch.distributed`` is initialised.  This switch only sets the per-layer opt-out the operator
    reads (``stp3_local_stats``); class, parameter names and state-dict keys stay the reference's."""
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.stp3_local_stats = not enabled
    return module


class GradientBuckets:
    """Flat fp32 parameter / gradient buckets in backward order, all-reduced while backward continues.

    ``gather=True`` (default): ``p.grad`` is reset to ``None`` before backward so autograd keeps the incoming
    gradient tensors as they are, and a finished bucket is filled with one multi-tensor copy (``finish()`` fills what
    the hooks have not); no per-parameter accumulation launches.
    ``gather=False``: every ``p.grad`` IS a view of its bucket and autograd accumulates into it in place (one small
    ``add_`` per parameter per step, ~480 launches for this model); same values (``0 + g`` vs ``g``)."""

    def __init__(self, module, bucket_bytes=8 << 20, process_group=None, average=True, gather=None):
        self.gather = True if gather is None else bool(gather)
        self.reductions_launched = 0                     # gradient-bucket all-reduces issued so far (bench.py's N > 1 line)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        from . import ops
        # buckets reduced from their hooks: more than one rank (or the one-rank exercise of that path, ops.FORCE_EXCHANGE)
        self.exchange = self.world > 1 or (dist.is_initialized() and ops.FORCE_EXCHANGE)
        self.average = average
        params = [p for p in module.parameters() if p.requires_grad]
        self.params = params[::-1]                       # backward order
        self.buckets = []                                # (flat gradient, [params])
        self.flat_params = []                            # flat parameter buffer per bucket
        self.grad_views = []                             # per bucket: the gradient view of every parameter
        self._bucket_of = {}
        cur, cur_bytes = [], 0
        for p in self.params:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._finished = False
        self._works = []
        self._hooks = []
        if self.exchange:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.broadcast_parameters(module)

    def _close(self, params):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        flat_param = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        off = 0
        views = []
        with torch.no_grad():
            for p in params:
                # Parameters AND gradients become views into the bucket's flat buffers, with the parameter's
                # own strides (channels-last weights keep their physical layout), so gradients accumulate
                # straight into the bucket and the optimizer runs on a handful of flat tensors.
                size, stride = tuple(p.size()), tuple(p.stride())
                view = torch.as_strided(flat_param, size, stride, off)
                view.copy_(p.data)
                p.data = view
                gview = torch.as_strided(flat, size, stride, off)
                views.append(gview)
                # an operator that produces this parameter's WHOLE gradient in one launch may write it here directly
                # (ops._conv2d_wgrad: no gather copy for the convolution weights, ~95 % of the gradient bytes)
                p._stp3_grad_view = gview
                # gather mode: autograd must find no gradient tensor on the parameter, or it accumulates in place into
                # the view and ``_gather`` cannot tell "already in the bucket" from "took no part in this step"
                p.grad = None if self.gather else gview
                off += p.numel()
                self._bucket_of[p] = len(self.buckets)
        self.buckets.append((flat, list(params)))
        self.grad_views.append(views)
        self.flat_params.append(flat_param)

    def broadcast_parameters(self, module):
        if self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=self.group)

    def zero_grad(self):
        if self.buckets[0][0].is_cuda:
            from . import ops
            ops.flush_wgrad_reductions()     # (a backward pass nobody finished: its partial sums must not outlive it)
        for p in self.params:
            p._stp3_uses = 0                 # applications of the weight in the coming pass (ops.note_weight_use)
        if self.buckets[0][0].is_cuda:
            ops.reset_weight_uses()
        if self.gather:
            for p in self.params:
                p.grad = None
        else:
            for flat, _ in self.buckets:
                flat.zero_()
        self._pending = [len(ps) for _, ps in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._finished = False

    @torch.no_grad()
    def _gather(self, i):
        """gather mode: move the gradients autograd left on the parameters of bucket ``i`` into its flat buffer
        (one multi-tensor copy) and point every ``p.grad`` back at its view."""
        flat, params = self.buckets[i]
        views = self.grad_views[i]
        # (a gradient that was written straight into its slice -- same memory, another tensor object -- needs no copy)
        have = [(v, p.grad) for v, p in zip(views, params)
                if p.grad is not None and p.grad is not v and p.grad.data_ptr() != v.data_ptr()]
        # p.grad is its view: a backward that ran without a preceding ``zero_grad()`` accumulated in place -- already in
        # the bucket, keep it.  p.grad is None: the parameter took no part in this step -- its slice must read zero.
        absent = [v for v, p in zip(views, params) if p.grad is None]
        if len(absent) == len(params):
            flat.zero_()
        elif absent:
            torch._foreach_zero_(absent)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(views, params):
            p.grad = v

    def _launch(self, i):
        self._launched[i] = True
        if self.gather:
            self._gather(i)
        if self.exchange:
            flat = self.buckets[i][0]
            self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.reductions_launched += 1

    def _on_grad(self, p):
        i = self._bucket_of[p]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def finish(self):
        """Call after ``backward``: completes buckets that were not launched from the hooks (single process, or
        parameters that received no gradient), waits for the in-flight all-reduces and averages over ranks.
        Idempotent within a step; the clipping / optimizer entry points call it themselves."""
        if self._finished:
            return
        self._finished = True
        if self.buckets[0][0].is_cuda:
            from . import ops
            ops.flush_wgrad_reductions()     # the deferred split-K sums of the weight gradients: one launch for all layers
        for i in range(len(self.buckets)):
            if not self._launched[i] and (self.gather or self.exchange):
                self._launch(i)
        if not self.exchange:
            return
        for w in self._works:
            w.wait()
        self._works = []
        if self.average:
            for flat, _ in self.buckets:
                flat.div_(self.world)

    def clip_grad_norm_(self, max_norm):
        """Global-norm clipping on the flat buckets (train.py:48 ``gradient_clip_val``)."""
        from .utils import staged_sum
        self.finish()
        total = torch.sqrt(torch.stack([staged_sum(f.float().square()) for f, _ in self.buckets]).sum())
        scale = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for flat, _ in self.buckets:
            flat.mul_(scale.to(flat.dtype))
        return total


class FlatAdam:
    """Adam (torch.optim.Adam semantics: L2 weight decay folded into the gradient, bias correction, no
    amsgrad; the reference's optimizer, trainer.py:456-462) applied to the flat parameter / gradient
    buffers of ``GradientBuckets``: a few elementwise kernels per bucket instead of several per parameter
    (the model has ~440 parameter tensors)."""

    def __init__(self, buckets, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.buckets = buckets
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        dev = buckets.flat_params[0].device
        # the step counter lives on the device so that a captured hipGraph of the step advances it on replay
        self.step_t = torch.zeros((), dtype=torch.float32, device=dev)
        self.exp_avg = [torch.zeros_like(p) for p in buckets.flat_params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in buckets.flat_params]
        self._table = None                                       # stp3_optim_bucket[] for the fused path

    @property
    def step_count(self):
        return int(self.step_t.item())

    @torch.no_grad()
    def step(self):
        self.buckets.finish()
        b1, b2 = self.betas
        self.step_t.add_(1.0)
        bc1 = 1.0 - torch.pow(b1, self.step_t)                      # bias corrections, device scalars
        bc2_sqrt = torch.sqrt(1.0 - torch.pow(b2, self.step_t))
        step_size = self.lr / bc1
        for (grad, _), param, m, v in zip(self.buckets.buckets, self.buckets.flat_params, self.exp_avg,
                                          self.exp_avg_sq):
            g = grad.add(param, alpha=self.weight_decay) if self.weight_decay else grad
            m.lerp_(g, 1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (v.sqrt() / bc2_sqrt).add_(self.eps)
            param.sub_((m / denom) * step_size)
        from . import ops
        ops.invalidate_weight_cache()        # the parameters are views of `param`: their version counters did not move
        ops.flush_batch_counters()           # STP3_LAZY_BN_COUNTER: num_batches_tracked of all layers, one launch

    def clip_and_step(self, max_norm):
        """Gradient-norm clipping followed by the Adam update (the tail of every training step).  Default: the
        two torch-operator methods above (CPU buckets: the gloo tests).  GPU buckets: three launches of
        ``stp3_optim_clip_adam`` for all buckets together.  Returns the total gradient norm."""
        self.buckets.finish()
        if not (FUSED_ADAM and self.buckets.flat_params[0].is_cuda):
            total = self.buckets.clip_grad_norm_(max_norm)
            self.step()
            return total
        import ctypes
        from . import _lib, ops
        if self._table is None:
            arr = (_lib.OptimBucket * len(self.buckets.buckets))()
            block = 0
            for rec, (grad, _), param, m, v in zip(arr, self.buckets.buckets, self.buckets.flat_params, self.exp_avg,
                                                   self.exp_avg_sq):
                assert grad.dtype == param.dtype == torch.float32 and grad.is_contiguous() and param.is_contiguous()
                rec.grad, rec.param, rec.exp_avg, rec.exp_avg_sq = (grad.data_ptr(), param.data_ptr(), m.data_ptr(),
                                                                    v.data_ptr())
                rec.numel, rec.first_block = grad.numel(), block
                block += (grad.numel() + 4095) // 4096
            dev = self.buckets.flat_params[0].device
            self._table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self._blocks = block
            need = ctypes.c_size_t()
            _lib.check(_lib.lib().stp3_optim_workspace_bytes(block, ctypes.byref(need)), 'stp3_optim_workspace_bytes')
            self._workspace = torch.empty(need.value, dtype=torch.uint8, device=dev)
            self._state = torch.zeros(5, dtype=torch.float32, device=dev)
            self._state[0:1].copy_(self.step_t.reshape(1))
            self.step_t = self._state[0]                         # one device-side step counter for both paths
        b1, b2 = self.betas
        _lib.check(_lib.lib().stp3_optim_clip_adam(
            ops._ptr(self._table), len(self.buckets.buckets), self._blocks, float(max_norm or 0.0), float(self.lr),
            float(b1), float(b2), float(self.eps), float(self.weight_decay), ops._ptr(self._state),
            ops._ptr(self._workspace), self._workspace.numel(), ops._stream()), 'stp3_optim_clip_adam')
        ops.invalidate_weight_cache()
        ops.flush_batch_counters()
        return self._state[4]

    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}
