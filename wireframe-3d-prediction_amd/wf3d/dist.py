"""One-process-per-GPU data parallelism for the wireframe path.

The path shards perfectly over clouds (LayerNorm, per-cloud pools, per-sample
attention: nothing crosses samples — SURVEY.md §8e), so the only exchange is the
sum of parameter gradients once per step.  `torch.distributed` backend "nccl"
is RCCL on ROCm; xGMI is point-to-point, so a ring all-reduce is bound by one
link (~153 GB/s): the 124 MB fp32 gradient set is cut into a few large buckets
in BACKWARD order (edge head -> vertex head -> encoder) and each bucket's
all-reduce is launched as soon as its last gradient has been accumulated, which
puts the edge+vertex buckets (63 % of the bytes) under the encoder backward
(~90 % of the step's compute).  The reference has no distributed code at all.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(device_type=None):
    """Initialise the default process group from torchrun's env (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR/PORT).  Returns (rank, world, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = (device_type or ("cuda" if torch.cuda.is_available() else "cpu")) == "cuda"
    if use_cuda:
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # "nccl" is RCCL on ROCm.  WF3D_DIST_BACKEND=gloo lets several ranks share one GPU (tests).
        backend = os.environ.get("WF3D_DIST_BACKEND", "nccl" if use_cuda else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def shard_batch(global_batch, rank, world):
    """Contiguous slice [lo, hi) of the global sample indices owned by `rank`."""
    per = global_batch // world
    if per * world != global_batch:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    return rank * per, (rank + 1) * per


@torch.no_grad()
def sync_parameters(module, src=0, group=None):
    """Broadcast every parameter/buffer from `src` (after lazily created layers —
    VertexPredictor.point_pool_proj, SURVEY.md §9 Q1 — exist on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def _stage_of(name):
    if name.startswith("edge_predictor."):
        return 0
    if name.startswith("vertex_predictor."):
        return 1
    if name.startswith("encoder.feature_fusion."):
        return 2
    # per-point MLP, last layer first.  Its gradients appear one layer at a time over the final 7 ms of the step, so each
    # layer of `encoder.mlp` (Linear + LayerNorm: indices 4k .. 4k+3) is its own bucket and leaves as soon as it is complete;
    # the two first layers (2 MB together) share the last one — what remains exposed after backward is that small reduce.
    if name.startswith("encoder.mlp."):
        try:
            layer = int(name.split(".")[2]) // 4
        except ValueError:
            return 3
        return 3 + (4 - max(layer, 1)) if layer <= 4 else 3      # reference depth: layers 0..4 -> stages 6, 6, 5, 4, 3
    return 3


class GradReducer:
    """Bucketed, overlapped gradient averaging.

    usage:   red = GradReducer(model);  loss.backward();  red.finish()      (any number of backward() calls per finish())

    Every rank issues the SAME sequence of collectives whatever its local gradients look like (a collective is matched by
    its position in that sequence, so a rank-local decision to issue one more, or to issue two in another order, hangs the
    job or pairs the wrong buffers):
      * early buckets hold the parameters that are known — collectively — to receive gradients, in backward order
        (edge head -> vertex head -> fusion -> per-point MLP).  A bucket's all-reduce is launched from a
        post-accumulate-grad hook once all its gradients have landed AND every earlier bucket has been launched;
        whatever was not launched by the end of backward is launched by finish(), in the same order, with zeros in
        place of a missing gradient.  A second gradient for a parameter whose bucket is already in flight (gradient
        accumulation: the program does that on every rank alike) marks the bucket for one more reduce in finish().
      * every other parameter (never seen with a gradient: EdgePredictor.spatial_proj; a lazily
        created layer in its first steps) sits in the TAIL bucket, reduced by finish() after backward — never stale —
        together with one flag per parameter, "this rank has a gradient".  The summed flags tell every rank the same
        thing: a count of `world` promotes the parameter to the early buckets (from the next step on), a count strictly
        between 0 and `world` is a rank-asymmetric gradient.
      * find_unused=False (default): the summed flags are copied to the host asynchronously and looked at in the NEXT
        finish() (no host sync in the step); a rank-asymmetric count raises there, on every rank at once.  Averages are
        written back wherever a local .grad exists (early buckets: always, a missing .grad is created).
      * find_unused=True: the flags are read in the same finish() (one small host sync per step) and the averaged value
        is written into .grad — created where it is None — for every parameter that ANY rank had a gradient for, so the
        ranks' optimizers stay in step even when their gradient sets differ.
    `exposed_ms()` = time the compute stream spent inside finish() (un-overlapped communication)."""

    def __init__(self, module, bucket_mb=48.0, group=None, average=True, find_unused=False):
        self.module, self.group, self.average, self.find_unused = module, group, average, find_unused
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self._buckets, self._tail = None, None
        self._hooks = []
        self._expected = set()           # parameters every rank has had a gradient for (learned from the summed flags)
        self._pending = None             # (host flags, event, tail params) of the previous finish(), default mode
        self._events = []
        self._build()
        if self.world > 1 and torch.cuda.is_available():
            # the reduces run beside the backward pass: cut the large wgrad launches into two rounds of workgroups so a
            # CU the collective's kernels hold costs them 15 % instead of 67 % (include/wf3d.h, wf3d_set_option)
            from . import _lib
            _lib.load().wf3d_set_option(b"tn_rounds", int(os.environ.get("WF3D_TN_ROUNDS", "2")))

    def _build(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad]
        order = sorted(range(len(named)), key=lambda i: (_stage_of(named[i][0]), -i))
        buckets, cur, cur_bytes, cur_stage = [], [], 0, None
        for i in order:
            n, p = named[i]
            if p not in self._expected:
                continue
            st = _stage_of(n)
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > self.bucket_bytes or st != cur_stage):
                buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_stage = st
        if cur:
            buckets.append(cur)
        self._buckets = []
        self._owner = {}
        for bi, ps in enumerate(buckets):
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
            self._buckets.append({"params": ps, "flat": flat, "got": set(), "work": None, "launched": False,
                                  "stale": False, "views": self._views(flat, ps)})
            for p in ps:
                self._owner[p] = bi
        tail = [named[i][1] for i in order if named[i][1] not in self._expected]
        if tail:
            flat = torch.zeros(sum(p.numel() for p in tail) + len(tail), dtype=tail[0].dtype, device=tail[0].device)
            self._tail = {"params": tail, "flat": flat, "work": None, "views": self._views(flat, tail),
                          "flags": flat[flat.numel() - len(tail):]}
        else:
            self._tail = None
        self._next = 0
        self._n_params = len(named)
        if self.world > 1:
            for _, p in named:
                if p in self._owner:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    @staticmethod
    def _views(flat, ps):
        views, off = [], 0
        for p in ps:
            n = p.numel()
            views.append(flat[off:off + n].view_as(p))
            off += n
        return views

    def _on_grad(self, p):
        b = self._buckets[self._owner[p]]
        if b["launched"] or p in b["got"]:
            b["stale"] = True                       # a second gradient (accumulation): one more reduce in finish()
        b["got"].add(p)
        self._launch_ready()

    def _launch_ready(self):
        # strictly in bucket order: the same sequence of collectives on every rank
        while self._next < len(self._buckets):
            b = self._buckets[self._next]
            if not all(q in b["got"] for q in b["params"]):
                return
            self._launch(b)
            self._next += 1

    def _launch(self, b):
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b["params"]]
        torch._foreach_copy_(b["views"], grads)
        b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        b["launched"] = True

    def _consume_pending(self):
        """Default mode: the summed tail flags of the PREVIOUS finish() (their device-to-host copy has long finished)."""
        if self._pending is None:
            return
        host, ev, params = self._pending
        self._pending = None
        if ev is not None:
            ev.synchronize()
        self._apply_flags(host.tolist(), params, strict=True)

    def _apply_flags(self, counts, params, strict):
        promote = []
        for c, p in zip(counts, params):
            c = int(round(c))
            if c == self.world or (c > 0 and not strict):
                promote.append(p)
            elif c != 0:
                raise RuntimeError(
                    f"wf3d.dist.GradReducer: a parameter of shape {tuple(p.shape)} received a gradient on {c} of {self.world} "
                    "ranks in the previous step.  The ranks' gradient sets differ; construct GradReducer(..., "
                    "find_unused=True), which reduces and writes back such gradients exactly (one small host sync per step).")
        if promote:
            self._expected.update(promote)
            self._dirty = True

    def finish(self):
        """Launch what the hooks have not (in order), redo buckets that accumulated a second gradient, reduce the tail
        bucket with its flags, wait, and write the averages into .grad."""
        if self.world == 1:
            return
        timed = torch.cuda.is_available() and any(p.is_cuda for p in self.module.parameters())
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self._dirty = False
        self._consume_pending()
        n = sum(1 for p in self.module.parameters() if p.requires_grad)
        if n != self._n_params:                     # a lazy parameter appeared: drain, re-bucket (it starts in the tail)
            for b in self._buckets:
                if b["work"] is not None:
                    b["work"].wait()
                    b["launched"] = False
            self._build()
        for b in self._buckets[self._next:]:        # first round, same order as the hooks would have used
            self._launch(b)
        self._next = len(self._buckets)
        for b in self._buckets:                     # accumulation: the early reduce carried a partial sum
            if b["stale"]:
                b["work"].wait()
                self._launch(b)
        t = self._tail
        if t is not None:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in t["params"]]
            torch._foreach_copy_(t["views"], grads)
            t["flags"].copy_(torch.tensor([0.0 if p.grad is None else 1.0 for p in t["params"]], dtype=t["flat"].dtype))
            t["work"] = dist.all_reduce(t["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        scale = 1.0 / self.world if self.average else 1.0
        for b in self._buckets:
            b["work"].wait()
            if scale != 1.0:
                b["flat"].mul_(scale)
            for p, v in zip(b["params"], b["views"]):
                if p.grad is None:
                    p.grad = v.clone()              # a parameter of the early buckets missed its gradient on this rank
            torch._foreach_copy_([p.grad for p in b["params"]], b["views"])
            b["got"], b["work"], b["launched"], b["stale"] = set(), None, False, False
        self._next = 0
        if t is not None:
            t["work"].wait()
            t["work"] = None
            nvals = t["flat"].numel() - len(t["params"])
            if scale != 1.0:
                t["flat"][:nvals].mul_(scale)
            if self.find_unused:
                counts = t["flags"].cpu().tolist()   # the one host sync of this mode
                for c, p, v in zip(counts, t["params"], t["views"]):
                    if c > 0.5:
                        if p.grad is None:
                            p.grad = v.clone()
                        else:
                            p.grad.copy_(v)
                self._apply_flags(counts, t["params"], strict=False)
            else:
                pairs = [(p.grad, v) for p, v in zip(t["params"], t["views"]) if p.grad is not None]
                if pairs:
                    torch._foreach_copy_([a for a, _ in pairs], [b_ for _, b_ in pairs])
                host = t["flags"].to("cpu", non_blocking=True)
                ev = None
                if t["flags"].is_cuda:
                    ev = torch.cuda.Event()
                    ev.record()
                self._pending = (host, ev, list(t["params"]))
        if self._dirty:
            self._build()                           # promoted parameters move to the early buckets from the next step on
        if timed:
            e1.record()
            self._events.append((e0, e1))

    def exposed_ms(self, reset=True):
        """Per-step times (ms) the compute stream spent in finish(): all-reduce not hidden under backward, plus the
        scale / copy-back kernels.  Synchronises."""
        if not self._events:
            return []
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._events]
        if reset:
            self._events = []
        return out

    def bucket_summary(self):
        out = [(len(b["params"]), b["flat"].numel() * b["flat"].element_size()) for b in self._buckets]
        if self._tail is not None:
            out.append((len(self._tail["params"]), self._tail["flat"].numel() * self._tail["flat"].element_size()))
        return out
