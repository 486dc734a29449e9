"""The training step as ONE hipGraph.

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
replays against the eager step bit for bit.

More than one rank (train.py:43-56: DDP + sync_batchnorm): the step then contains RCCL collectives -- the BatchNorm statistics
exchanges on the sequential chain of layers and the gradient-bucket all-reduces issued from the hooks during backward -- and
they are captured WITH it (RCCL kernels are graph nodes like any other; ``torch.distributed`` enqueues them on its own stream
and joins it to the capturing stream with events, which the capture records as cross-stream edges).  Nothing in the step
changes: the same operators issue the same collectives in the same order, so a rank that replays and a rank that (after a
failed capture) launches eagerly still meet in every collective; ``bench.py`` nevertheless makes the ranks agree on one mode.
What the capture buys at N > 1 is what it buys at N = 1 -- the 31-39 ms of host work per step disappear -- plus the ~200
blocking host-side collective calls of the statistics exchanges.  Exercised on hardware with ONE RCCL rank taking the N > 1 code
path (``ops.FORCE_EXCHANGE``, tests/test_graph_exchange_gpu.py: RCCL refuses two ranks on one device); no multi-GPU node
was available to any round, so no scaling curve has been measured.
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
        if buckets.exchange:
            # The process group's watchdog thread polls the events of the collectives the warm-up steps enqueued until it has
            # seen them complete (one pass every 100 ms); a poll that lands inside the capture fails with
            # hipErrorCapturedEvent and takes the process down (3 of 3 runs, profiles/r06c_graph_exchange_capture.txt).  The
            # device is idle here, so a second is ample for the watchdog to retire everything -- and collectives recorded
            # DURING a capture are never handed to it.
            import time
            time.sleep(1.0)
        ops.flush_batch_counters()
        self.graph = torch.cuda.CUDAGraph()
        # (with collectives in the step the process group's watchdog thread polls events of EARLIER work while this thread
        # captures: `relaxed` keeps that legal -- the default mode fails the capture on any such call from any thread; ROCm 7.2
        # does not start a `thread_local` capture at all: capture_begin trips over an inactive capture status, r06c)
        mode = 'relaxed' if buckets.exchange else 'global'
        exchanges0, reductions0 = ops.exchange_counts()['batchnorm'], buckets.reductions_launched
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode=mode):
            self.loss = self._body()
        torch.cuda.synchronize(dev)
        # the collectives ONE replay issues (counted where they were recorded: the host does not see them again)
        self.collectives = {'batchnorm_statistics_all_reduces': ops.exchange_counts()['batchnorm'] - exchanges0,
                            'gradient_bucket_all_reduces': buckets.reductions_launched - reductions0}
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
