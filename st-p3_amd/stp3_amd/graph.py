"""The training step as ONE hipGraph.

At B=4 per GPU the perception step is ~3 500 kernel launches of 5-100 us each: launched eagerly the
host (Python dispatch + autograd) cannot keep the GPU fed (profiles/: GPU busy ~65 %).  Every shape
in the step is static -- the voxel pool never syncs with the host (dropped points carry id -1,
SURVEY.md section 7 item 14) -- so forward, backward, gradient clipping and the Adam update are
captured once into a hipGraph and replayed per batch.  Per batch the host only
  1. copies the batch into the graph's static input buffers (skipped for tensors that already are
     those buffers),
  2. rebuilds the geometry-only pooling plan into its static buffers (pose math on the host, a few
     hundred floats uploaded, index kernels),
  3. replays the graph.
Single-process only: with more than one rank the step contains RCCL collectives (gradient buckets,
cross-replica BatchNorm) and stays eager.
"""
import torch

_POSE_KEYS = ('intrinsics', 'extrinsics', 'future_egomotion')


class GraphedTrainStep:
    def __init__(self, module, buckets, optimizer, grad_clip, batch, autocast_dtype=torch.bfloat16, warmup=3,
                 log=lambda msg: None):
        self.module, self.buckets, self.optimizer = module, buckets, optimizer
        self.grad_clip, self.autocast_dtype = grad_clip, autocast_dtype
        self.model = module.model
        from . import ops
        ops.WEIGHT_CACHE_ENABLED = False     # weight casts must be captured into (and replayed with) the graph
        dev = next(module.parameters()).device
        self.device = dev
        # static inputs: device tensors are used in place (the caller may keep writing new batches into them)
        self.static = {}
        for k, v in batch.items():
            self.static[k] = v if (not torch.is_tensor(v) or v.is_cuda) else v
        self.static_dev = {k: v for k, v in self.static.items() if torch.is_tensor(v) and v.is_cuda}
        # the training step consumes the ego-motion on the device as well (label warps, ego-motion planes)
        self.ego_dev = batch['future_egomotion'].to(dev).clone()
        self.plan = self.model.prepare_plan(*(batch[k] for k in _POSE_KEYS), dev)
        self.loss = None
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for i in range(warmup):
                self._body()
                torch.cuda.synchronize(dev)
                log(f'graph: eager warm-up {i} done')
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()
        torch.cuda.synchronize(dev)
        log('graph: captured')

    def _graph_batch(self):
        b = dict(self.static)
        b['future_egomotion'] = self.ego_dev
        return b

    def _body(self):
        self.buckets.zero_grad()
        with torch.autocast('cuda', dtype=self.autocast_dtype):
            loss = self.module.training_step(self._graph_batch())
        loss.backward()
        self.buckets.finish()
        self.buckets.clip_grad_norm_(self.grad_clip)
        self.optimizer.step()
        return loss.detach()

    def __call__(self, batch):
        # One stream synchronisation per step, before the plan buffers of the previous replay are rewritten.
        # Back-to-back replays with the plan rebuild enqueued in between, host running several steps ahead,
        # fault on ROCm 7.2 ("write access to a read-only page"); with the host at most one step ahead they
        # do not.  The host work per step is ~1 ms, so this costs a few percent at most.
        torch.cuda.current_stream(self.device).synchronize()
        for k, dst in self.static_dev.items():
            src = batch[k]
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        self.ego_dev.copy_(batch['future_egomotion'], non_blocking=True)
        self.model.prepare_plan(*(batch[k] for k in _POSE_KEYS), self.device, out=self.plan)
        self.graph.replay()
        return self.loss
