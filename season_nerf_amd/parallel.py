"""Ray-parallel execution over the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

Rays are independent given the weights, so the path shards with NO collective inside the data path:
every rank holds a replica of the 3.3 MB weights, renders a contiguous block of rays, and the only exchange is one
all-gather of the rendered rows (RGB tiles: 12 B/ray) - latency-bound, a few tens of microseconds over xGMI.
"""
import collections

import torch
import torch.distributed as dist

# Collectives this process has issued on the hot path, by kind - so that a run under torch.distributed (also with ONE rank: RCCL
# initialised, stream-ordered, timed, proving nothing about scaling) can show which exchanges really executed (bench.py `collectives`).
COLLECTIVES = collections.Counter()


def data_parallel(group=None):
    """True where the data-parallel exchanges run: an initialised process group THIS rank is a member of - a world of ONE rank
    included, so that the RCCL code path can be exercised on a single GPU (tests/test_gpu_rccl_world1.py); a rank outside
    `group` never enters its collectives."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return group is None or dist.get_rank(group) >= 0


def global_min(local_min, group=None):
    """(minimum over all ranks, world size) of a small detached tensor: ONE `ReduceOp.MIN` all-reduce (RCCL, stream-ordered).
    Used by `training.get_loss` for the `SK_Albedo` term, whose minimum the reference takes over the WHOLE batch
    (Eval_Tools_2.py:374).  Without a process group: (local_min, 1)."""
    if not data_parallel(group):
        return local_min, 1
    g = local_min.detach().clone()
    dist.all_reduce(g, op=dist.ReduceOp.MIN, group=group)
    COLLECTIVES["albedo_min_all_reduce"] += 1
    return g, dist.get_world_size(group)


def shard_bounds(n, world):
    """Contiguous, balanced ray ranges: the first n % world ranks get one extra ray."""
    q, r = divmod(n, world)
    out, lo = [], 0
    for k in range(world):
        hi = lo + q + (1 if k < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


def shard_dict(data_dict, rank, world):
    """Slice every per-ray tensor of a reference-style data dict (Top, Bot, Sun_Angle, Time_Encoded, ...)."""
    n = data_dict["Top"].shape[0]
    lo, hi = shard_bounds(n, world)[rank]
    return {k: (v[lo:hi] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v) for k, v in data_dict.items()}


def gather_rows(local, n_total, group=None):
    """All-gather row blocks of unequal length (shard_bounds order) into the full [n_total, ...] tensor on every rank.
    One collective: blocks are padded to the largest shard."""
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_total, world)
    mx = max(hi - lo for lo, hi in bounds)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    COLLECTIVES["rows_all_gather"] += 1
    return torch.cat([buf[k * mx: k * mx + (hi - lo)] for k, (lo, hi) in enumerate(bounds)], 0)


class TileGroupGather:
    """Asynchronous all-gather of per-step result tiles, `group` steps per collective (fewer, larger collectives: one 48 KB RGB
    tile per step is pure launch latency - measured 10 % of a 1.4 ms render step against 1.3 % with 8 steps per gather).
    Tile groups are double-buffered: the gather of one group overlaps the kernels that fill the next.

        tg = TileGroupGather((R, 3), group=8, device=dev)
        for i in range(steps):
            out = tg.slot()            # [R, 3] view: let the kernel of step i write its tile here (current stream)
            ...launch...
            tg.commit()                # starts the gather when the group is full
        tg.flush()                     # gathers a last, partly filled group and waits for everything
        tiles = tg.gathered(k)         # [world, R, 3] of step k (valid for the last two groups)
    """

    def __init__(self, tile_shape, group=8, device=None, dtype=torch.float32, pg=None):
        self.pg, self.G = pg, int(group)
        self.world = dist.get_world_size(pg) if dist.is_initialized() else 1
        self.tiles = [torch.empty((self.G,) + tuple(tile_shape), dtype=dtype, device=device) for _ in range(2)]
        self.full = [torch.empty((self.world, self.G) + tuple(tile_shape), dtype=dtype, device=device) for _ in range(2)]
        self.pending = [None, None]
        self.n = 0

    def slot(self):
        b, k = (self.n // self.G) & 1, self.n % self.G
        if k == 0 and self.pending[b] is not None:
            self.pending[b].wait()                 # the gather that last read this tile group (two groups ago)
            self.pending[b] = None
        return self.tiles[b][k]

    def _start(self, b):
        if dist.is_initialized():
            flat = self.full[b].view((self.world * self.G,) + tuple(self.tiles[b].shape[1:]))      # concatenation form: every backend takes it
            self.pending[b] = dist.all_gather_into_tensor(flat, self.tiles[b], group=self.pg, async_op=True)
            COLLECTIVES["tile_group_all_gather"] += 1
        else:
            self.full[b][0].copy_(self.tiles[b])

    def commit(self):
        b, k = (self.n // self.G) & 1, self.n % self.G
        self.n += 1
        if k == self.G - 1:
            self._start(b)

    def flush(self):
        if self.n % self.G:
            self._start((self.n // self.G) & 1)
        for b in (0, 1):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None

    def reset(self):
        self.flush()
        self.n = 0

    def gathered(self, step):
        """[world, *tile_shape] of `step` (after the gather of its group has been waited for, e.g. by flush())."""
        b, k = (step // self.G) & 1, step % self.G
        return self.full[b][:, k]


class ShardedEval:
    """Wraps an evaluator's `eval`: each rank evaluates its ray shard, per-ray results are all-gathered."""

    def __init__(self, eval_tool, network, group=None, keys=("Rendered_Col",)):
        self.eval_tool, self.network, self.group, self.keys = eval_tool, network, group, keys

    def eval(self, data_dict, current_step, train_mode):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        n = data_dict["Top"].shape[0]
        out = self.eval_tool.eval(shard_dict(data_dict, rank, world), self.network, current_step, train_mode)
        return {k: gather_rows(out[k].contiguous(), n, self.group) for k in self.keys}


def allreduce_gradients(module_or_params, group=None, average=True):
    """Data-parallel training: sum (average) the gradients of all ranks in ONE flat bucket (3.3 MB at W=256, 12.8 MB at
    W=512 - a single RCCL all-reduce over xGMI).  Works with any optimiser; parameters without a gradient on
    this rank contribute zeros.  Each rank's loss is a mean over its own rays, so `average=True` reproduces the
    global-batch mean when shards are equal.  Note: train-mode BatchNorm statistics stay per rank."""
    params = list(module_or_params.parameters()) if hasattr(module_or_params, "parameters") else list(module_or_params)
    params = [p for p in params if p.requires_grad]
    if not params or not data_parallel(group):
        return
    store = getattr(module_or_params, "_param_store", None)
    if store is not None and all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(store.param_list, store.grad_views)):
        # a season_nerf_amd.T_NeRF after a training backward: every .grad is a view of one flat arena - reduce it in place
        dist.all_reduce(store.grads, op=dist.ReduceOp.SUM, group=group)
        COLLECTIVES["grad_arena_all_reduce"] += 1
        if average:
            store.grads /= dist.get_world_size(group)
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    COLLECTIVES["grad_bucket_all_reduce"] += 1
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
