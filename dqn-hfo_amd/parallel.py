"""Data-parallel form of the update: one process per GPU.

Two transports:
  * NATIVE (default on GPUs): RCCL inside libdqnhip.so (include/dqnhip.h dqnhip_dp_*): the two
    gradient all-reduces are ncclAllReduce calls on the learner's own stream, issued by
    dqnhip_dp_update — no Python between the phases.  torch.distributed is only the rendezvous
    channel that carries rank 0's 128-byte RCCL id to the other ranks (a C++ host uses
    dqnhip_dp_init_file or its own launcher instead) — make_native_data_parallel().
  * torch.distributed all-reduce between dqnhip_update_phase calls (DataParallelUpdate): what the
    CPU/gloo tests drive with the oracle as the backend, and the fallback transport when two
    ranks must share one GPU (RCCL refuses duplicate devices).

The reference has no collective at all (threads + a mutex, src/dqn_main.cpp:62-63,
359-363); this is the MI355X-native addition of SURVEY.md §8e.  The update has exactly
two exchange points — after the critic backward and after the actor backward — because
the actor step reads the UPDATED critic (src/dqn.cpp:904 -> 914).  Each rank gathers its
own minibatch slice from its own replay shard, so no data-path collective exists; the
EuclideanLoss normaliser is the GLOBAL batch, the actor gradient is an un-normalised sum
(src/dqn.cpp:918-921), hence a plain sum all-reduce reproduces the single-GPU update up
to summation order.  The clip norm is taken on the reduced gradient, so every rank
applies the identical Adam step and the replicas never diverge.

`backend` is anything with update_phase(phase, idx) — the HIP learner on the GPU, a
stand-in in the CPU/gloo tests.
"""
import os

import torch
import torch.distributed as dist


class DataParallelUpdate:
    def __init__(self, backend, critic_grad, actor_grad, group=None, overlap=False):
        self.backend = backend
        self.critic_grad = critic_grad      # flat view: critic gradient arena + [loss_sum, q_sum, 0, 0]
        self.actor_grad = actor_grad
        self.group = group
        # The online actor's forward mu(s) (src/dqn.cpp:910-911) neither reads nor writes the critic
        # gradients: a backend that can split it off (phase 10 / 11 of dqnhip_update_phase) runs it
        # while the first all-reduce is in flight on RCCL's own stream.  OFF by default: measured on one
        # rank the split costs +18 us (the two actors' layers no longer share launches) and the async
        # collective +35 us more (cross-stream edges), against <= 36 us of forward it can hide.
        self.overlap = bool(overlap) and getattr(backend, "supports_split_phase0", False)

    def update(self, idx=None):
        b = self.backend
        if self.overlap:
            b.update_phase(10, idx)
            work = dist.all_reduce(self.critic_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            b.update_phase(11, None)
            work.wait()                     # stream dependency, not a host sync
        else:
            b.update_phase(0, idx)
            dist.all_reduce(self.critic_grad, op=dist.ReduceOp.SUM, group=self.group)
        b.update_phase(1, None)
        dist.all_reduce(self.actor_grad, op=dist.ReduceOp.SUM, group=self.group)
        b.update_phase(2, None)


def make_hip_data_parallel(pkg, state_size, rank, world, device, group=None, overlap=False, **dqn_kwargs):
    """Build a HIP learner whose gradient arenas live in a torch tensor (so RCCL can reduce
    them in place) and which enqueues on torch's current stream (so collectives and kernels
    are ordered by the stream, no host sync)."""
    torch.cuda.set_device(device)
    hidden = dqn_kwargs.get("hidden", (1024, 512, 256, 128))
    B = dqn_kwargs.get("minibatch", 32)
    nbytes = pkg.DQN.grad_arena_bytes(state_size, B, hidden, dqn_kwargs.get("precision", "fp32"))
    arena = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda:%d" % device)
    stream = torch.cuda.current_stream().cuda_stream
    dqn = pkg.DQN(state_size, device=device, dp_world=world, dp_rank=rank, stream=stream,
                  grad_arena=arena.data_ptr(), grad_arena_bytes=nbytes, **dqn_kwargs)
    views = []
    for net in (pkg.ACTOR, pkg.CRITIC):
        ptr, n = dqn.grad_buffer(net)
        off = (ptr - arena.data_ptr()) // 4
        views.append(arena[off:off + n])
    dp = DataParallelUpdate(dqn, critic_grad=views[1], actor_grad=views[0], group=group,
                            overlap=overlap)
    dp.arena = arena       # keep the storage alive
    if world > 1:
        sync_params_from_rank0(dqn, group)
    return dqn, dp


def sync_params_from_rank0(dqn, group=None):
    """Replicas must start identical: broadcast rank 0's weights (4 nets), Adam history and
    iterations (torch.distributed transport; the native path does this inside dqnhip_dp_init)."""
    import numpy as np
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    for net in range(4):
        for kind in ((0, 1, 2) if net < 2 else (0,)):
            t = torch.from_numpy(dqn.get_params(net, kind)).to(dev)
            dist.broadcast(t, src=0, group=group)
            dqn.set_params(net, t.cpu().numpy(), kind)
    it = torch.tensor([dqn.actor_iter(), dqn.critic_iter()], dtype=torch.int64, device=dev)
    dist.broadcast(it, src=0, group=group)
    dqn.set_iters(int(it[0]), int(it[1]))


class NativeDataParallel:
    """dqnhip_dp_update: phase 0 -> ncclAllReduce(critic grads) -> phase 1 -> ncclAllReduce(actor
    grads) -> phase 2, enqueued by the library on the learner's stream."""
    overlap = False

    def __init__(self, dqn):
        self.backend = dqn

    def update(self, idx=None):
        self.backend.dp_update(idx)

    def update_n(self, n):
        self.backend.dp_update_n(n)


def make_native_data_parallel(pkg, state_size, rank, world, device, group=None, per_layer=False, half_grads=False,
                              shard_opt=False, unverified_ok=False, **dqn_kwargs):
    """One learner per rank with an RCCL communicator inside the library.  Needs an initialised
    torch.distributed group only to ship the id when world > 1.  use_graph=True (a DQN keyword) makes
    dqnhip_dp_update replay the whole update, collectives included, as one captured hipGraph; shard_opt=True shards
    the optimiser over the group (DQNHIP_DP_SHARD_OPT: reduce-scatter / Adam on 1/N / all-gather)."""
    dqn = pkg.DQN(state_size, device=device, dp_world=world, dp_rank=rank, **dqn_kwargs)
    box = [pkg.DQN.dp_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    dqn.dp_init(box[0], per_layer=per_layer, half_grads=half_grads, shard_opt=shard_opt, unverified_ok=unverified_ok)
    return dqn, NativeDataParallel(dqn)
