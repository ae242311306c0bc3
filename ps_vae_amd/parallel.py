"""Data-parallel training over the graphs of a batch: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).

Exact global-batch semantics.  The three loss terms are means over different
counts (non-pad target tokens, selected pairs, graphs), so averaging per-rank
gradients would NOT equal the gradient of the global batch.  Instead every rank
divides its local loss SUMS by the GLOBAL counts (the 4-float `denoms` vector of
the batch is all-reduced before it is used) and the flat gradient buffers are
all-reduced with SUM.  The result is independent of how graphs are dealt to
ranks, up to fp32 summation order.

The reference has no working multi-GPU path for this model (its Lightning 'dp'
mode mis-splits the batch dict, `src/train.py:49-50`), so this file has no
reference counterpart beyond "train on more GPUs".
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous slice of a global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class GradReducer:
    """All-reduce (sum) of a flat gradient buffer.

    `bounds` are bucket boundaries (element offsets); each bucket is one
    collective.  xGMI is a point-to-point mesh and the whole gradient is only
    ~7.6 MB, so the transfer is latency-bound: as long as the collectives run
    after backward (not overlapped with it) ONE bucket `[0, n]` is the fastest
    choice; `model.bucket_bounds` lists where backward finishes groups of
    parameters, for callers that overlap (`OverlappedReducer`, the default of `make_reducer`)."""

    def __init__(self, bounds, group=None, always=False):
        self.bounds = list(bounds)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # `always`: issue the collectives even in a group of one (rehearses the RCCL path on a single GPU)
        self.active = dist.is_initialized() and (self.world > 1 or always)
        # gloo moves GPU tensors through host memory anyway; doing that copy here keeps gloo on its CPU path
        self.stage_on_host = dist.is_initialized() and dist.get_backend(group) == 'gloo'

    def _all_reduce(self, t, async_op=False):
        if self.stage_on_host and t.is_cuda:
            host = t.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def reduce_denoms(self, local_denoms, out=None, async_op=False):
        """Global loss denominators: sum of the per-rank counts.  With `async_op` the collective is only
        enqueued (it depends on nothing but the batch, so it can run a step ahead); the returned handle's
        `wait()` makes the current stream wait for it.  Returns `out`, or `(out, handle)`."""
        out = local_denoms.clone() if out is None else out.copy_(local_denoms)
        work = self._all_reduce(out, async_op=async_op) if self.active else None
        return (out, work) if async_op else out

    def reduce_grads(self, flat_grads, async_op=False):
        if not self.active:
            return []
        works = []
        for lo, hi in zip(self.bounds[:-1], self.bounds[1:]):
            if hi > lo:
                work = self._all_reduce(flat_grads[lo:hi], async_op=async_op)
                if work is not None:
                    works.append(work)
        return works if async_op else []


class OverlappedReducer(GradReducer):
    """Gradient all-reduce in three messages in the order backward finishes them, the first two overlapped
    with the rest of backward.  The model's backward records an event once the piece branch's gradients (flat
    range `[0, b1)`, 1.0 MB) are final and another once those of the property head, link scorer, latent head and
    encoder.linear (`[b1, b3)`, 4.5 MB) are - both while the GINE stacks' backward, more than half of the
    backward pass, has not started; each range is reduced on a side stream as soon as its event fires, and the
    rest (`[b3, n)`, 2.1 MB: GINE stacks, node_trans, embedding tables) after backward.  xGMI moves such
    messages in a few tens of microseconds each, so the point of the split is to hide the first 5.5 MB behind
    compute, not to pipeline the wire.  `tail` extra elements behind the gradients (see
    `PSVAEModel.grad_message`) travel with the last message.

        red = OverlappedReducer(model)
        loss.backward(); red.reduce(); opt.step()

    The result is bit-identical to one all-reduce over the whole buffer (element-wise sums, same ranks)."""

    def __init__(self, model, group=None, always=False, tail=0):
        grads = model.grad_message()[0]
        b = model.bucket_bounds
        super().__init__([b[0], b[1], b[3], b[4] + int(tail)], group, always)
        if b[4] + int(tail) > grads.numel():
            raise ValueError('tail exceeds the spare floats behind the gradient buffer')
        self.model, self.message = model, grads
        self.events = model.early_gradient_events()
        self.comm = torch.cuda.Stream(device=grads.device) if grads.is_cuda else None

    def reduce(self):
        """Call after `backward`; returns when the three collectives are queued and the current stream waits
        for them."""
        if not self.active:
            return
        model = self.model
        grads = model.grad_message()[0]
        b0, b1, b2, b3 = self.bounds
        cur = torch.cuda.current_stream()
        if not model.last_backward_signalled:
            # an accumulating backward (or one before the events existed) signals nothing: everything is final
            # only once the current stream gets here
            self.comm.wait_stream(cur)
        works = []
        for event, lo, hi in ((self.events[0], b0, b1), (self.events[1], b1, b2)):
            if model.last_backward_signalled:
                self.comm.wait_event(event)
            with torch.cuda.stream(self.comm):
                works.append(self._all_reduce(grads[lo:hi], async_op=True))
        works.append(self._all_reduce(grads[b2:b3], async_op=True))      # queued behind backward on the current stream
        for w in works:
            if w is not None:
                w.wait()            # the current stream waits for the collective (it runs on the backend's own stream)
        cur.wait_stream(self.comm)  # and for whatever the side stream itself carried (host-staged copies)


def make_reducer(model, group=None, always=False, overlap=None):
    """The reducer a data-parallel loop should use: three messages overlapped with backward when the model lives
    on a GPU (`overlap=False` selects the single message after backward)."""
    model.flat_buffers()
    if dist.is_available() and dist.is_initialized():
        model.shard_rank = dist.get_rank(group)      # keys the epsilon rows of batches that carry no graph_base
    if overlap is None:
        overlap = True
    if overlap and model.flat_buffers()[1].is_cuda:
        return OverlappedReducer(model, group, always)
    return GradReducer([0, model._n_trained], group, always)


def global_step_reference(loss_sums_per_rank, counts_per_rank):
    """Host arithmetic of the scheme above (used by the CPU tests): per-rank loss sums and
    counts -> the global mean each rank's scaled losses add up to."""
    total = sum(counts_per_rank)
    return sum(s / total for s in loss_sums_per_rank)
