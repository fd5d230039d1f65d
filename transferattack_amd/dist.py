"""One process per GPU: how the 1000-image job is spread over the GPUs of a node.

The reference is single-process, single-GPU (main.py:31, 43); nothing here has a counterpart to port.

* Single-surrogate attacks shard by WHOLE reference batches (``shard_batches``): DIM draws one geometry per
  batch and Admix mixes images within a batch (dim.py:54-63, admix.py:44), so batches are never re-cut.  Every
  batch is seeded from (base_seed, batch_idx) (``seed_batch``) so the output does not depend on the GPU count.
  No collective on the data path.
* A list of surrogates (ENS, or any attack run on ``--model a,b,...``) puts one surrogate per rank of a model group:
  ``ShardedEnsemble.forward`` averages the logits with one RCCL all-reduce ([N,1000] fp32) and, in its backward, sums
  the members' input gradients with another ([N,3,224,224] fp32) -- the only two exchange steps the path has.  Ranks of a group see the same images and,
  after the second all-reduce, hold identical gradients, so each runs the identical fused update locally.
* Attacks that address single members (SVRE, CWA, AdaEA, SMER: ``self.model.models[k](x)``, svre.py:72-83,
  cwa.py:71-81, adaea.py:65-82, smer.py:80-106) use ``ShardedMembers``: ``models[k]`` is a handle that runs on the rank
  owning member k and broadcasts its logits forward and its input gradient backward.  Every rank of the group executes
  the same attack code on the same images with the same host draws (``seed_batch`` seeds torch AND numpy), so the
  results equal the single-device run bit for bit; what is sharded is the surrogates' weights and activations
  (one model per 288 GB GPU), not the arithmetic.  AdaEA, which evaluates ALL members at the same point, additionally
  gets one-round forms (``member_logits`` / ``member_input_grads`` / ``member_losses``: one all-gather, or all-gather +
  all-reduce, per round) so that M GPUs do the M evaluations side by side.
"""
import os
import random

import numpy as np
import torch
from torch.autograd.function import once_differentiable
import torch.distributed as dist
import torch.nn as nn


def _invalidate_partials(*tensors):
    """Drop the |g| tile sums a HIP kernel attached to a gradient that a collective has just replaced or modified in place
    (transferattack_amd._hip: the sums travel as an attribute of the gradient tensor)."""
    from . import _hip
    _hip.invalidate_partials(*tensors)


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, single_rank_group=False):
    """Join the job torch.distributed.run started (nccl == RCCL on ROCm; gloo for the CPU tests).  A world of one rank
    needs no process group and gets none, unless ``single_rank_group`` asks for it (RCCL smoke test on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world <= 1 and not single_rank_group) or dist.is_initialized():
        return rank_world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank_world()


def shard_batches(num_batches, rank, world):
    """Indices of the reference batches rank ``rank`` of ``world`` processes: round-robin, so the short tail batch
    (1000 = 31 x 32 + 8) lands on the rank with the fewest full batches and every rank's share differs by <= 1."""
    return list(range(rank, num_batches, world))


def seed_batch(base_seed, batch_idx):
    """Seed the three host generators the attacks draw from for one batch -- torch (DIM / Admix / SIA), numpy (SVRE /
    SMER member order) and Python's ``random`` (BSR's shuffles, L2T, OPS: bsr.py:44-60, l2t.py:500, ops.py:137) -- so
    (base_seed, batch_idx) -> the draws of that batch are the same whichever rank processes it, and all ranks of a model
    group apply the identical transform."""
    seed = (int(base_seed) * 1000003 + int(batch_idx) * 7919 + 12345) % (2 ** 63 - 1)
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))          # SVRE / SMER draw their member order from numpy (svre.py:75, smer.py:75)
    return seed


def model_groups(world, group_size):
    """Partition the ranks into groups of ``group_size`` consecutive ranks (one ensemble member per rank);
    returns (my_group, my_index_in_group, image_shard_index, num_image_shards).  Must be called by every rank."""
    rank, _ = rank_world()
    assert world % group_size == 0, "world size must be a multiple of the ensemble size"
    mine = None
    for first in range(0, world, group_size):
        ranks = list(range(first, first + group_size))
        grp = dist.new_group(ranks)
        if rank in ranks:
            mine = (grp, rank - first, first // group_size, world // group_size)
    return mine


class _AllReduceMean(torch.autograd.Function):
    """z = (1/M) sum_m z_m over the group; d(loss)/d z_m = (1/M) d(loss)/d z."""

    @staticmethod
    def forward(ctx, logits, group, members):
        out = logits.detach().clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        ctx.members = members
        return out / members

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        return grad / ctx.members, None, None


class _SumInputGrad(torch.autograd.Function):
    """identity on x; backward: all-reduce(sum) of d(loss)/dx over the group -- every member contributes
    (1/M) J_m^T dL/dz, the sum is what ``EnsembleModel`` (utils.py:82-105) gives on one device.  The sum goes into a
    fresh tensor and the producer-side |g| sums are dropped (they describe the local member's gradient only)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        total = grad.contiguous().clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=ctx.group)
        _invalidate_partials(total, grad)
        return total, None


class _AllGatherStack(torch.autograd.Function):
    """out[m] = member m's logits (all-gather over the group, member order = rank order within the group); every rank
    evaluates the same loss on the same stack, so d(loss)/d(own logits) is that rank's slice of the stack's gradient."""

    @staticmethod
    def forward(ctx, logits, group, members, index):
        parts = [torch.empty_like(logits) for _ in range(members)]
        dist.all_gather(parts, logits.detach().contiguous(), group=group)
        ctx.index = index
        return torch.stack(parts, dim=0)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        return grad[ctx.index].contiguous(), None, None, None


class ShardedEnsemble(nn.Module):
    """Drop-in for ``EnsembleModel`` (utils.py:82-105) when each rank of ``group`` holds ONE member: forward = local member +
    all-reduce(mean) of the logits for mode 'mean' (utils.py:100), all-gather into the [M, N, classes] stack for mode 'ind'
    (utils.py:101-103); the input gradient is the members' sum either way.  ``models`` / ``num_models`` / ``device`` /
    ``mode`` keep the attribute contract of the reference class."""

    def __init__(self, local_model, group, members, mode='mean'):
        super().__init__()
        self.local = local_model
        self.models = [local_model]
        self.group = group
        self.num_models = members
        self.mode = mode
        self.type_name = 'ensemble'
        self.device = next(local_model.parameters()).device
        self._index = None                      # this member's place in the stack: only mode 'ind' needs it (resolved lazily,
                                                # so constructing the module needs no process group)

    @property
    def index(self):
        if self._index is None:
            self._index = dist.get_group_rank(self.group, dist.get_rank()) if self.group is not None else dist.get_rank()
        return self._index

    def forward(self, x):
        if x.requires_grad:
            x = _SumInputGrad.apply(x, self.group)
        if self.mode == 'mean':
            return _AllReduceMean.apply(self.local(x), self.group, self.num_models)
        if self.mode == 'ind':
            return _AllGatherStack.apply(self.local(x), self.group, self.num_models, self.index)
        raise NotImplementedError


class _OwnerCall(torch.autograd.Function):
    """logits = member(x) where the member lives on one rank of the group.  Forward: the owner runs it and broadcasts
    the logits ([N, classes] fp32).  Backward: the owner back-propagates the (identical on every rank) logit gradient
    and broadcasts d/dx ([N,3,H,W] fp32).  The owner keeps its graph until the handle's output dies, so a second
    backward through the same logits (adaea.py:67 then :81) works like retain_graph=True."""

    @staticmethod
    def forward(ctx, x, handle):
        ctx.handle, ctx.own, ctx.x_shape = handle, None, tuple(x.shape)
        if handle.local is not None:
            if ctx.needs_input_grad[0]:
                with torch.enable_grad():
                    leaf = x.detach().requires_grad_(True)
                    out = handle.local(leaf)
                ctx.own = (leaf, out)
            else:
                out = handle.local(x.detach())
            logits = out.detach().clone().contiguous()
            handle.announce(logits)
        else:
            logits = x.new_empty((x.shape[0],) + handle.announce(None))
        dist.broadcast(logits, src=handle.owner, group=handle.group)
        return logits

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_logits):
        handle = ctx.handle
        if ctx.own is not None:
            leaf, out = ctx.own
            gx = torch.autograd.grad(out, leaf, grad_logits.contiguous(), retain_graph=True)[0].contiguous()
        else:
            gx = torch.empty(ctx.x_shape, dtype=grad_logits.dtype, device=grad_logits.device)
        dist.broadcast(gx, src=handle.owner, group=handle.group)
        _invalidate_partials(gx)                  # every rank of the group must take the same path through the update
        return gx, None


class MemberHandle(nn.Module):
    """``EnsembleModel.models[k]`` when member k lives on rank ``owner`` (a GLOBAL rank) of ``group``; ``local`` is the
    wrapped surrogate on the owner and None elsewhere.  Calls are collective: every rank of the group makes them in the
    same order (they do -- the ranks run the same attack on the same batch with the same seeds)."""

    def __init__(self, local, owner, group):
        super().__init__()
        self.local, self.owner, self.group = local, owner, group
        self._tail = None                        # logits shape after the batch axis, learnt at the first call

    def announce(self, logits):
        if self._tail is None:
            box = [tuple(logits.shape[1:])] if logits is not None else [None]
            dist.broadcast_object_list(box, src=self.owner, group=self.group)
            self._tail = tuple(box[0])
        return self._tail

    def forward(self, x):
        return _OwnerCall.apply(x, self)


class _GatherLogits(torch.autograd.Function):
    """All members at the same input in ONE round: every rank runs its own member, the logits are all-gathered
    ([M, N, classes]); backward: every rank back-propagates its own slice of the cotangent and the input gradients
    are summed by one all-reduce (the members' contributions to d/dx add up)."""

    @staticmethod
    def forward(ctx, x, owner):
        with torch.enable_grad():
            leaf = x.detach().requires_grad_(True)
            out = owner.local(leaf)
        ctx.own, ctx.owner = (leaf, out), owner
        owner._last_graph = (leaf, out)                 # member_input_grads differentiates this same forward
        mine = out.detach().contiguous()
        parts = [torch.empty_like(mine) for _ in range(owner.num_models)]
        dist.all_gather(parts, mine, group=owner.group)
        return torch.stack(parts, dim=0)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        leaf, out = ctx.own
        owner = ctx.owner
        gx = torch.autograd.grad(out, leaf, grad[owner.index].contiguous(), retain_graph=True)[0].contiguous()
        dist.all_reduce(gx, op=dist.ReduceOp.SUM, group=owner.group)
        _invalidate_partials(gx)                  # gx was modified in place behind torch's back
        return gx, None


class ShardedMembers(nn.Module):
    """Drop-in for ``EnsembleModel`` (utils.py:82-105) with one member per rank of ``group`` and the members
    individually addressable.  ``forward`` is the same stack-and-mean as the reference class, over the handles."""

    def __init__(self, local_model, index, group, group_ranks, mode='mean'):
        super().__init__()
        self.local = local_model
        self.group = group
        self.index = index
        self._last_graph = None
        self.models = [MemberHandle(local_model if k == index else None, r, group) for k, r in enumerate(group_ranks)]
        self.num_models = len(group_ranks)
        self.mode = mode
        self.type_name = 'ensemble'
        self.device = next(local_model.parameters()).device

    def forward(self, x):
        outputs = torch.stack([member(x) for member in self.models], dim=0)
        if self.mode == 'mean':
            return torch.mean(outputs, dim=0)
        if self.mode == 'ind':
            return outputs
        raise NotImplementedError

    # ---- one-round forms for algorithms that evaluate ALL members at the same point (AdaEA, adaea.py:65-82): M GPUs do
    # the M evaluations side by side instead of M owner broadcasts one after the other
    def member_logits(self, x):
        """[M, N, classes]: every member's logits at x, differentiable with respect to x"""
        return _GatherLogits.apply(x, self)

    def member_input_grads(self, loss_of_logits):
        """[d loss_of_logits(logits_m) / dx for m < M] at the x of the latest ``member_logits`` call, with ONE backward per
        rank through the forward that call already did: each rank differentiates its own member's loss, the M input
        gradients are all-gathered.  Each gradient is exactly what a single device computes for that member."""
        leaf, out = self._last_graph
        with torch.enable_grad():
            mine = torch.autograd.grad(loss_of_logits(out), leaf, retain_graph=True)[0].contiguous()
        parts = [torch.empty_like(mine) for _ in range(self.num_models)]
        dist.all_gather(parts, mine, group=self.group)
        _invalidate_partials(mine, *parts)
        return parts

    def member_losses(self, inputs, loss_of_logits):
        """[M, len(inputs)] matrix L[m][j] = loss_of_logits(member_m(inputs[j])), no gradient: every rank evaluates its
        member on all inputs, one all-gather of the rows"""
        with torch.no_grad():
            row = torch.stack([loss_of_logits(self.local(v)) for v in inputs]).contiguous()
        rows = [torch.empty_like(row) for _ in range(self.num_models)]
        dist.all_gather(rows, row, group=self.group)
        return torch.stack(rows, dim=0)

    def eval(self):
        self.local.eval()
        return super().eval()

    def parameters(self, recurse=True):
        return self.local.parameters(recurse)


PER_MEMBER_ATTACKS = ('svre', 'cwa', 'adaea', 'smer')        # these index self.model.models[k] (svre.py:72-83, ...)


def sharded_attack(cls, attack_name, model_names, world, **ctor_kwargs):
    """One surrogate per rank of a model group: build ``cls`` so that this rank loads only ITS member and sees the others
    through ``ShardedEnsemble`` (logit / input-gradient all-reduce) or, for the attacks that address single members,
    ``ShardedMembers``.  Collective: every rank calls it.  Returns (attacker, member_index, shard_rank, shard_world);
    images are sharded over the ``shard_world`` groups, the ranks of one group process the same batches."""
    members = len(model_names)
    grp, member, shard_rank, shard_world = model_groups(world, members)
    name = model_names[member]
    first = shard_rank * members
    group_ranks = list(range(first, first + members))
    per_member = attack_name in PER_MEMBER_ATTACKS

    class Sharded(cls):
        def load_model(self, model_name):
            local = super().load_model(name)
            if per_member:
                return ShardedMembers(local, member, grp, group_ranks)
            return ShardedEnsemble(local, grp, members)

    Sharded.__name__ = "Sharded" + cls.__name__
    attacker = Sharded(model_name=list(model_names) if per_member else name, **ctor_kwargs)
    return attacker, member, shard_rank, shard_world
