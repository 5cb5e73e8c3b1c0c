"""Data-parallel training step: one graph per GPU, flat gradient buckets, RCCL all-reduce overlapped with the backward.

The reference trains on one GPU with ``accumulate_grad_batches: 8`` (configs/tracking_cfg.yaml:4,
scripts/train.py:76); W GPUs x 1 graph with gradient averaging is the same optimisation step executed
in space instead of time.  The graphs share nothing but the weights, so the
data path has no collective; the only exchange is the all-reduce(sum) / W of the flat fp32 gradient buffer
per optimizer step (1.2 MB at the reference dims, 19 MB at d = 128), issued as two buckets: the gradients of the
message-passing modules and the classifier are final when the backward's last weight-gradient group is (on the
library's side stream) -- their all-reduce starts there and runs under the encoder's backward; the encoder's bucket
follows on the caller's stream."""
import torch

from . import capi
from .autograd import native_backward, native_forward_saved


def backward_available():
    """True when libmpnhip.so carries the hand-written backward."""
    lib = capi.load()
    return lib.mpnhip_backward_workspace_bytes(None, 0, 0) != 0


def shard_indices(n_items, rank, world_size):
    """Round-robin shard of sample (graph / sequence) ids: rank r takes r, r+W, r+2W, ..."""
    return list(range(rank, n_items, world_size))


def bce_logits_grad(logits, labels, first_step=0):
    """d/dlogits of the reference loss (pl_module.py:88-120): sum over the classified steps of
    BCEWithLogits(pos_weight = #neg/#pos), mean over edges.  logits [L,E], labels [E] in {0,1}."""
    E = labels.numel()
    pos = labels.sum()
    pw = (E - pos) / pos.clamp(min=1)
    w = torch.where(labels > 0, pw, torch.ones_like(pw))
    g = (torch.sigmoid(logits) * (1 + (pw - 1) * labels) - pw * labels) / max(E, 1)
    if first_step > 0:
        g[:first_step] = 0
    return g, w


def allreduce_mean_(flat, world_size, group=None):
    """In-place all-reduce(sum) / W of a flat gradient bucket (RCCL over xGMI on GPUs, gloo in CPU tests)."""
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world_size)
    return flat


class FlatBucket:
    """Parameters AND their gradients as views of two flat fp32 buffers: one collective and one optimizer kernel per
    step.  The parameters' storage is moved into ``flat_params`` (values preserved; ``state_dict`` unaffected)."""

    def __init__(self, params):
        self.params = list(params)
        # every parameter starts on a 16-byte boundary of the flat buffers (padding stays zero: zero gradient, zero update), so
        # that the kernels' 16-byte operand paths apply to the views exactly as to separately allocated tensors
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        dev = self.params[0].device
        # four spare elements behind the gradients: [n] carries a rank's "invalid graph" flag through the all-reduce (TrainStep)
        self.n = n
        self.flat = torch.zeros(n + 4, dtype=torch.float32, device=dev)
        self.flat_params = torch.zeros(n + 4, dtype=torch.float32, device=dev)
        self.views = {}
        self.offsets = dict((id(p), off) for p, off in zip(self.params, offs))
        for p, off in zip(self.params, offs):
            v = self.flat[off:off + p.numel()].view_as(p)
            self.views[id(p)] = v
            pv = self.flat_params[off:off + p.numel()].view_as(p)
            pv.copy_(p.data)
            p.data = pv
            p.grad = v

    def zero_(self):
        self.flat.zero_()


class FlatAdam:
    """``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (the reference's optimizer, pl_module.py:76-77) as ONE
    native kernel over a ``FlatBucket``'s flat parameter / gradient buffers (``mpnhip_adam_step``)."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.bucket = bucket
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(bucket.flat_params)
        self.exp_avg_sq = torch.zeros_like(bucket.flat_params)
        self.t = 0   # calls of step()
        # guarded calls that the device-side flag skipped: kept ON the device next to the moments (the host never reads the flag on
        # the ranks whose own graph is fine), so that the bias corrections follow the number of APPLIED updates like torch's Adam
        self.skipped = torch.zeros(4, dtype=torch.int32, device=bucket.flat_params.device)

    @property
    def applied_steps(self):
        """Updates actually applied (a host read of the device-side counter; diagnostics / tests)."""
        return self.t - int(self.skipped[0])

    def step(self, guarded=False):
        """``guarded``: skip the update (on the device, no host read) when the bucket's spare element ``flat[n]`` is non-zero."""
        import ctypes as C
        self.t += 1
        capi._weights_epoch[0] += 1   # raw-pointer update: invalidates packed weight images kept by MOTMPNet.hot_path
        b = self.bucket
        skip = C.c_void_p(b.flat.data_ptr() + 4 * b.n) if guarded else None
        capi.check(capi.load().mpnhip_adam_step_counted(capi.ptr(b.flat_params), capi.ptr(b.flat), capi.ptr(self.exp_avg),
                                                        capi.ptr(self.exp_avg_sq), b.n, C.c_float(self.lr),
                                                        C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                                        C.c_float(self.weight_decay), self.t, skip,
                                                        capi.ptr(self.skipped) if guarded else None, capi.stream_ptr()),
                   "mpnhip_adam_step")


class TrainStep:
    """fwd (saving activations) -> loss gradient -> hand-written bwd into the flat bucket ->
    all-reduce(sum)/W over RCCL (world_size > 1) -> Adam.  Graph prep is cached on `holder`."""

    def __init__(self, model, world_size=1, lr=1e-3, process_group=None, weight_decay=0.0, force_collectives=False):
        """``force_collectives``: issue the step's collectives (and the guarded optimizer step) even at world_size 1 -- a 1-rank
        process group exercises the whole data-parallel code path (RCCL communicator, the side stream as an ExternalStream, two
        asynchronous bucket all-reduces, the join) where only one GPU is available; the result is the single-rank step's."""
        if not backward_available():
            raise capi.MpnhipError("TrainStep needs mpnhip_backward")
        self.model = model
        self.world_size = world_size
        self.collectives = world_size > 1 or bool(force_collectives)
        self.pg = process_group
        self.bucket = FlatBucket(model.hot_path_parameters())
        # hot_path_parameters(): encoder.node_model, encoder.edge_model, then the message-passing modules and the classifier:
        # flat[:enc_elems] is the encoder's bucket (final last, on the caller's stream), flat[enc_elems:] the other one
        enc = set(id(p) for p in list(model.encoder.node_model.parameters()) + list(model.encoder.edge_model.parameters()))
        first_mp = [p for p in self.bucket.params if id(p) not in enc][0]
        self.enc_elems = self.bucket.offsets[id(first_mp)]
        self._side = None
        self.opt = FlatAdam(self.bucket, lr=lr, weight_decay=weight_decay)
        self.first_class_step = max(int(model.num_enc_steps) - int(model.num_class_steps), 0)
        self._cmodels = {}

    def native_models(self, n_edges):
        """The native model descriptions (``mpnhip_model``: every weight's and gradient's device address) of the forward and of the
        backward, built once per resolved operand precision and reused: the parameters are views of the bucket's flat buffer and
        the gradients views of its gradient buffer, so the addresses do not change from step to step.  Building one costs 50 - 70 us
        of Python, a step needs several, and at the reference's graph sizes the step is bound by the host (KITTIMOTS-like graph:
        453 us to enqueue a 390 us step, tools/diag/host_cost.py).  The cache is keyed on the parameters' addresses and shapes,
        ``num_enc_steps``, the reattach flags and the aggregation mode."""
        model = self.model
        prec = model.operand_precision(n_edges)
        # everything c_model() bakes into the description besides the precision: the parameters' addresses and shapes (layer
        # dims), the step count, the reattach flags and the aggregation -- a model edited between two steps gets a new description
        agg = getattr(model.MPNet.node_model, "node_agg_fn", None)
        ptrs = (tuple((p.data_ptr(), tuple(p.shape)) for p in self.bucket.params), int(model.num_enc_steps),
                bool(model.reattach_initial_nodes), bool(model.reattach_initial_edges), getattr(agg, "code", id(agg)))
        hit = self._cmodels.get(prec)
        if hit is None or hit[0] != ptrs:
            keep_f, keep_b = [], []
            mf = model.c_model(keep_f, n_edges=n_edges)
            mb = model.c_model(keep_b, grads=self.bucket.views, n_edges=n_edges)
            hit = (ptrs, mf, mb, keep_f, keep_b)
            self._cmodels[prec] = hit
        return hit[1], hit[2]

    def allreduce_buckets(self, side_pending):
        """all-reduce(sum) / W of the flat gradient buffer.  ``side_pending``: the backward left its side stream un-joined
        (MPNHIP_BWD_DEFER_SIDE_JOIN): the message-passing bucket's collective is ordered behind THAT stream, so it overlaps
        the encoder's backward still queued on the caller's stream; the encoder's bucket follows in the caller's order."""
        import torch.distributed as dist
        flat, k = self.bucket.flat, self.enc_elems
        with torch.cuda.device(flat.device):
            if side_pending:
                if self._side is None:
                    self._side = torch.cuda.ExternalStream(capi.load().mpnhip_side_stream(), device=flat.device)
                # evidence of the overlap (tests, bench): when the side stream reached the message-passing bucket's collective, and
                # when the caller's stream finished the encoder's backward
                self.ev_mp_ready = torch.cuda.Event(enable_timing=True)
                self.ev_main_done = torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(self._side):
                    self.ev_mp_ready.record()
                    w_mp = dist.all_reduce(flat[k:], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                self.ev_main_done.record()
                w_enc = dist.all_reduce(flat[:k], op=dist.ReduceOp.SUM, group=self.pg, async_op=True) if k else None
                w_mp.wait()
                if w_enc is not None:
                    w_enc.wait()
                capi.check(capi.load().mpnhip_side_stream_join(capi.stream_ptr()), "mpnhip_side_stream_join")
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)
            flat.mul_(1.0 / self.world_size)

    def __call__(self, x, edge_index, edge_attr, labels=None, holder=None, optimizer_step=True, edge_graph=None, n_graphs=1):
        """``edge_graph`` / ``n_graphs``: (x, edge_index, edge_attr) is a block-diagonal batch of ``n_graphs`` graphs (int32 graph id per
        edge, ``loss.edge_graph_ids``): ONE forward / backward over the batch with the reference's per-graph loss averaged over the
        graphs -- ``accumulate_grad_batches`` optimizer micro-steps (configs/tracking_cfg.yaml:3-4) executed in space on one GPU."""
        from .mpn import _prepared, check_hot_path_inputs
        model = self.model
        g = _prepared(edge_index, x.shape[0], holder)
        x = capi.f32c(x)
        ea = capi.f32c(edge_attr)
        E = ea.shape[0]
        m_fwd, m_bwd = self.native_models(E)
        check_hot_path_inputs(m_fwd, g, x, ea)
        L = max(int(model.num_enc_steps), 1)
        logits = torch.empty((L, E), dtype=torch.float32, device=x.device)
        ws = native_forward_saved(model, g, x, ea, logits, cmodel=m_fwd)
        if labels is None:
            if getattr(self, "_default_labels", None) is None or self._default_labels.numel() != E:
                self._default_labels = (torch.arange(E, device=x.device) % 7 == 0).float()
            labels = self._default_labels
        # reference loss (pl_module.py:88-107) and its gradient w.r.t. every step's logits, natively
        from .loss import tracking_loss_and_grad
        self.last_loss, glog = tracking_loss_and_grad(logits, labels, self.first_class_step, 1.0, edge_graph=edge_graph, n_graphs=n_graphs)
        self.bucket.zero_()
        lib = capi.load()
        defer = self.collectives and bool(lib.mpnhip_backward_uses_side_stream(m_fwd))
        native_backward(model, g, x, ea, glog, ws, self.bucket.views, defer_side_join=defer, cmodel=m_bwd)
        # IndexError like the reference's x[row] gather for an edge_index outside [0, N) (graph prep clamps such entries and sets a
        # flag): read once per graph, BEFORE anything is done with the gradients of the clamped graph -- the prep finished long
        # before the backward was enqueued, so the read waits for nothing
        err = None
        try:
            g.raise_if_invalid()
        except IndexError as exc:
            err = exc
        if self.collectives:
            # every rank must take part in this step's collectives; the flag rides in a spare element of the bucket, the reduced
            # element guards the optimizer step on every rank (nobody steps on bad gradients), then the offending rank raises
            if err is not None:
                self.bucket.flat[self.bucket.n:].fill_(1.0)
                torch.cuda.current_stream().synchronize()   # (the bucket's collective may run on the library's side stream)
            self.allreduce_buckets(defer)
        elif err is not None:
            raise err
        if optimizer_step:
            self.opt.step(guarded=self.collectives)
        if err is not None:
            raise err
        return logits
