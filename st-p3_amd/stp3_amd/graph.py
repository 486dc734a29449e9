"""The training step as ONE hipGraph (single process).

A step of BASELINE configs[2] is ~1 900 kernel launches of 5-100 us; launched eagerly the Python side of the step costs
31-39 ms against 39-40 ms of GPU time: the host is the wall the kernels run into.  Every shape in the
step is static, no operator reads a value back, every scratch buffer is owned by an operator -- so forward, backward,
gradient gather, clipping, Adam and the refresh of the bf16 weight shadows are captured ONCE and replayed per batch.  The
captured step is a single stream (with a prepared plan the forward forks no side stream): a multi-stream form of the step --
the encoder's two heads on two streams, the weight gradients of leaf parameters beside the data gradients -- was built and
measured this round: about a millisecond faster launched eagerly, 2.4 ms SLOWER captured (hipGraph turns every cross-stream
edge into a barrier between hardware queues: 41.0 vs 38.5 ms, profiles/r05h_*), and it is deleted: replays of the single-stream
capture run the ~1 900 kernels back to back (99 % of the wall time in kernels).

Per batch the host does what depends on the batch and nothing else (``TrainingModule.prepare_batch``): copies the batch into
the graph's static input buffers (skipped for tensors that already are those buffers), rebuilds the geometry-only voxel-pool
plan and the label-warp matrices into their static buffers (pose mathematics on the host, a few hundred floats uploaded,
five index kernels), counts the BatchNorm batch counters, replays.

Semantics kept (reference: stp3/trainer.py:101-172 ``shared_step``, :456-462 Adam, train.py:48 gradient clipping): the
captured body is ``bench.py``'s eager step, operator for operator; tests/test_graph_step_gpu.py pins the loss trajectory of
replays against the eager step bit for bit.  With more than one rank the step contains RCCL collectives and stays eager.
"""
import torch

from . import ops

_POSE_KEYS = ('intrinsics', 'extrinsics', 'future_egomotion')


class GraphedTrainStep:
    def __init__(self, module, buckets, optimizer, grad_clip, batch, autocast_dtype=torch.bfloat16, warmup=3,
                 log=lambda msg: None):
        self.module, self.buckets, self.optimizer = module, buckets, optimizer
        self.grad_clip, self.autocast_dtype = grad_clip, autocast_dtype
        dev = next(module.parameters()).device
        assert dev.type == 'cuda', 'a hipGraph step needs a GPU'
        self.device = dev
        # static inputs: device tensors are used in place (the caller may keep writing new batches into them); the pose
        # tensors live on the host (ops.lift_matrices builds the bit-exact geometry constants there)
        self.static = {k: v for k, v in batch.items() if not (torch.is_tensor(v) and k in _POSE_KEYS)}
        self.static_dev = {k: v for k, v in self.static.items() if torch.is_tensor(v) and v.is_cuda}
        self.poses = (batch['intrinsics'], batch['extrinsics'])
        self.prepared = module.prepare_batch(batch, dev)
        self.loss = None
        # Autograd keeps a parameter's AccumulateGrad node -- and the stream that was current when it was made -- alive for
        # as long as any graph that reaches the parameter is alive (e.g. an un-detached loss of an earlier eager step): such
        # a node would run on ITS stream, outside the capture.  Drop what the collector can drop before the warm-up builds
        # fresh nodes under the capture stream.
        import gc
        gc.collect()
        self.stream = torch.cuda.Stream(device=dev)            # warm-up AND capture: scratch buffers are per stream
        cur = torch.cuda.current_stream(dev)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for i in range(warmup):
                self._body()
                log(f'graph: eager warm-up {i} done')
        cur.wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        ops.flush_batch_counters()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.loss = self._body()
        torch.cuda.synchronize(dev)
        # capturing records the kernels, it does not run them: take back what the host counted during the capture pass
        self.counted = ops.pending_batch_counters()            # the BatchNorm layers one step runs in training mode
        ops.discard_pending_batch_counters()
        module.training_step_count -= 1
        self.replays = 0
        log(f'graph: captured ({len(self.counted)} BatchNorm counters per step)')

    def _graph_batch(self):
        b = dict(self.static)
        b['future_egomotion'] = self.prepared['ego']           # (device: the temporal model's ego-motion bias reads it there)
        b['intrinsics'], b['extrinsics'] = self.poses          # (host; the plan is prepared: forward only slices them)
        b['_prepared'] = self.prepared
        return b

    def _body(self):
        self.buckets.zero_grad()
        model = self.module.model
        model.prebuilt_plan = self.prepared['plan']            # forward pools with the prepared plan (no host work) ...
        try:
            with torch.autocast('cuda', dtype=self.autocast_dtype):
                loss = self.module.training_step(self._graph_batch())
        finally:
            model.prebuilt_plan = None                         # ... and only this step does
        loss.backward()
        self.buckets.finish()
        self.optimizer.clip_and_step(self.grad_clip)
        return loss.detach()

    def __call__(self, batch=None):
        """One training step on ``batch`` (None: the batch the graph was captured on, e.g. bench.py's resident batch)."""
        if batch is not None:
            for k, dst in self.static_dev.items():
                src = batch[k]
                if src is not dst:
                    dst.copy_(src, non_blocking=True)
            self.poses = (batch['intrinsics'], batch['extrinsics'])
            self.module.prepare_batch(batch, self.device, out=self.prepared)
        self.module.model.prebuilt_plan = None
        self.graph.replay()
        ops.count_batches_again(self.counted)
        self.module.training_step_count += 1
        self.replays += 1
        return self.loss
