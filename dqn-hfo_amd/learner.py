"""Host-side mirror of the reference's `dqn::DQN` (src/dqn.hpp:56-134) for the hot
path, over the C-ABI of include/dqnhip.h.  Method names, argument meaning and error
behaviour follow the reference; where the reference aborts through glog CHECK /
LOG(FATAL) this raises `DQNFatal`.

PyTorch is not used here at all — device memory and streams live inside the native
library; torch only appears in parallel.py (torch.distributed over RCCL).
"""
import ctypes as C
import os
from collections import namedtuple

import numpy as np

from . import capi
from .capi import ACTOR, CRITIC, ACTOR_TARGET, CRITIC_TARGET, KIND_W, KIND_M, KIND_V, KIND_G

kActionSize = 4          # src/dqn.hpp:20
kActionParamSize = 6     # src/dqn.hpp:21
kStateInputCount = 1     # src/dqn.hpp:18
DASH, TURN, TACKLE, KICK = 0, 1, 2, 3   # hfo::action_t values pinned by src/dqn.cpp:181-186

Action = namedtuple("Action", "action arg1 arg2")          # src/hfo_game.hpp:7-11
Transition = namedtuple("Transition", "state actor_output reward on_policy_target next_state")
# next_state is None for terminal transitions (boost::none, src/dqn.cpp:878)


class DQNFatal(RuntimeError):
    """The reference's CHECK/LOG(FATAL) abort, as an exception."""


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(capi.fp)


def GetParamOffset(action, arg_num=0):
    """src/dqn.cpp:162-178."""
    if arg_num < 0 or arg_num > 1:
        return -1
    if action == DASH:
        return arg_num
    if action == TURN:
        return 2 if arg_num == 0 else -1
    if action == TACKLE:
        return 3 if arg_num == 0 else -1
    if action == KICK:
        return 4 + arg_num
    raise DQNFatal("Unrecognized action: %r" % (action,))


def GetAction(actor_output):
    """src/dqn.cpp:196-208: argmax over the 4 logits with TACKLE masked to -99999
    (std::max_element: the first maximum wins), then the chosen action's parameters."""
    ao = np.asarray(actor_output, dtype=np.float32)
    copy = ao[:kActionSize].copy()
    copy[TACKLE] = -99999
    act = int(np.argmax(copy))                 # np.argmax also returns the first maximum
    o1 = GetParamOffset(act, 0)
    assert o1 >= 0
    o2 = GetParamOffset(act, 1)
    return Action(act, float(ao[kActionSize + o1]), 0.0 if o2 < 0 else float(ao[kActionSize + o2]))


def PrintActorOutput(ao):
    """src/dqn.cpp:210-216."""
    f = lambda v: "%f" % v
    return ("Dash(" + f(ao[4]) + ", " + f(ao[5]) + ")=" + f(ao[0]) + ", Turn(" + f(ao[6]) + ")=" + f(ao[1])
            + ", Tackle(" + f(ao[7]) + ")=" + f(ao[2]) + ", Kick(" + f(ao[8]) + ", " + f(ao[9]) + ")=" + f(ao[3]))


class DQN:
    supports_split_phase0 = True        # dqnhip_update_phase accepts 10 / 11 (parallel.DataParallelUpdate)
    """Device-resident learner with the reference's method surface.

    Constructor arguments replace the reference's SolverParameter pair + gflags
    (src/dqn.cpp:21-31, src/dqn_main.cpp:249-262); defaults are the reference's.
    """

    def __init__(self, state_size, minibatch=32, hidden=(1024, 512, 256, 128), memory=500000,
                 gamma=0.99, beta=0.5, tau=0.001, soft_update_freq=1, actor_lr=1e-5, critic_lr=1e-3,
                 momentum=0.95, momentum2=0.999, clip_grad=10.0, memory_threshold=1000, seed=1,
                 device=0, dp_world=1, dp_rank=0, use_graph=False, stream=None, grad_arena=None,
                 grad_arena_bytes=0, tid=0, save_path="state/dqn", precision="fp32", loss_scale=0.0, tuning=0):
        self.lib = capi.load()
        cfg = capi.Config()
        self.lib.dqnhip_default_config(C.byref(cfg), state_size)
        cfg.minibatch = minibatch
        cfg.num_hidden = len(hidden)
        for i in range(capi.MAX_HIDDEN):
            cfg.hidden[i] = hidden[i] if i < len(hidden) else 0
        cfg.replay_capacity = memory
        cfg.soft_update_freq = soft_update_freq
        cfg.gamma, cfg.beta, cfg.tau = gamma, beta, tau
        cfg.actor_lr, cfg.critic_lr = actor_lr, critic_lr
        cfg.momentum, cfg.momentum2, cfg.clip_gradients = momentum, momentum2, clip_grad
        cfg.device, cfg.dp_world, cfg.dp_rank, cfg.use_graph = device, dp_world, dp_rank, int(use_graph)
        cfg.seed = seed
        cfg.stream = stream
        cfg.grad_arena = grad_arena
        cfg.grad_arena_bytes = grad_arena_bytes
        cfg.precision = {"fp32": 0, "fp16": 1}[precision]
        cfg.loss_scale = loss_scale
        cfg.tuning_flags = int(tuning)                  # capi.TUNE_* bits: alternative schedules of the same arithmetic
        self.cfg = cfg
        self.h = capi.H()
        self._ck(self.lib.dqnhip_create(C.byref(cfg), C.byref(self.h)))
        self.state_size_ = state_size
        self.kMinibatchSize = minibatch
        self.memory_threshold = memory_threshold        # FLAGS_memory_threshold, src/dqn.cpp:26
        self.gamma_ = gamma
        self.tid_ = tid
        self.save_path_ = save_path
        self.unum_ = 0
        self.random_engine = np.random.default_rng(seed)   # stands in for std::mt19937
        self.smoothed_critic_loss_ = 0.0
        self.smoothed_actor_loss_ = 0.0
        self.last_snapshot_iter_ = 0
        self.snapshot_freq = 10000                      # FLAGS_snapshot_freq, src/dqn.cpp:28
        self.select_actions_cap = 0                     # 0: the reference's cap (the minibatch, src/dqn.cpp:699); -1: none; n: n

    # -- plumbing -----------------------------------------------------------------
    @staticmethod
    def grad_arena_bytes(state_size, minibatch, hidden, precision="fp32"):
        lib = capi.load()
        cfg = capi.Config()
        lib.dqnhip_default_config(C.byref(cfg), state_size)
        cfg.minibatch = minibatch
        cfg.precision = {"fp32": 0, "fp16": 1}[precision]
        cfg.num_hidden = len(hidden)
        for i, hsz in enumerate(hidden):
            cfg.hidden[i] = hsz
        return lib.dqnhip_grad_arena_bytes(C.byref(cfg))

    def _ck(self, rc):
        if rc != 0:
            raise DQNFatal(self.lib.dqnhip_last_error().decode())

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self._ck(self.lib.dqnhip_destroy(self.h))      # refuses while sharers exist
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- accessors (src/dqn.hpp:112, 127-134) ------------------------------------------
    def memory_size(self):
        n = C.c_int32()
        self._ck(self.lib.dqnhip_memory_size(self.h, C.byref(n)))
        return n.value

    def _iters(self):
        a, c = C.c_int32(), C.c_int32()
        self._ck(self.lib.dqnhip_get_iters(self.h, C.byref(a), C.byref(c)))
        return a.value, c.value

    def actor_iter(self):
        return self._iters()[0]

    def critic_iter(self):
        return self._iters()[1]

    def max_iter(self):
        return max(self._iters())

    def min_iter(self):
        return min(self._iters())

    def set_iters(self, actor_iter, critic_iter):
        self._ck(self.lib.dqnhip_set_iters(self.h, actor_iter, critic_iter))

    def state_size(self):
        return self.state_size_

    def save_path(self):
        return self.save_path_

    def unum(self):
        return self.unum_

    def set_unum(self, unum):
        self.unum_ = unum

    # -- replay memory -------------------------------------------------------------------
    def ClearReplayMemory(self):
        self._ck(self.lib.dqnhip_clear_memory(self.h))

    def LabelTransitions(self, transitions):
        """src/dqn.cpp:783-797: fills on_policy_target by the reverse discounted scan;
        returns the relabelled list (the reference mutates in place)."""
        if len(transitions) == 0:
            raise DQNFatal("Need at least one transition to label.")
        r = _f32([t.reward for t in transitions])
        mc = np.empty_like(r)
        self._ck(self.lib.dqnhip_label_transitions(self.gamma_, _p(r), r.size, _p(mc)))
        return [t._replace(on_policy_target=float(m)) for t, m in zip(transitions, mc)]

    def _pack(self, transitions):
        S = self.state_size_
        n = len(transitions)
        s = np.empty((n, S), np.float32); nx = np.zeros((n, S), np.float32)
        a = np.empty((n, 10), np.float32); r = np.empty(n, np.float32); mc = np.empty(n, np.float32)
        term = np.zeros(n, np.uint8)
        for i, t in enumerate(transitions):
            if len(t.state) != S:
                raise DQNFatal("state size %d != %d" % (len(t.state), S))
            s[i] = t.state; a[i] = t.actor_output; r[i] = t.reward; mc[i] = t.on_policy_target
            if t.next_state is None:
                term[i] = 1
            else:
                nx[i] = t.next_state
        return s, a, r, mc, nx, term

    def AddTransition(self, transition):
        """src/dqn.cpp:768-773."""
        s, a, r, mc, nx, term = self._pack([transition])
        self._ck(self.lib.dqnhip_add_transition(self.h, _p(s), _p(a), float(r[0]), float(mc[0]), _p(nx), int(term[0])))

    def AddTransitions(self, transitions):
        """src/dqn.cpp:775-781."""
        s, a, r, mc, nx, term = self._pack(transitions)
        self.add_transitions_arrays(s, a, r, mc, nx, term)

    def add_transitions_arrays(self, s, a, r, mc, nx, term):
        """AddTransitions on already-packed arrays ([n,S], [n,10], [n], [n], [n,S], [n] u8)."""
        s, a, r, mc, nx = _f32(s), _f32(a), _f32(r), _f32(mc), _f32(nx)
        term = np.ascontiguousarray(term, dtype=np.uint8)
        self._ck(self.lib.dqnhip_add_transitions(self.h, _p(s), _p(a), _p(r), _p(mc), _p(nx),
                                                 term.ctypes.data_as(capi.up), r.size))

    def read_memory(self, first, n):
        S = self.state_size_
        s = np.empty((n, S), np.float32); nx = np.empty((n, S), np.float32)
        a = np.empty((n, 10), np.float32); r = np.empty(n, np.float32); mc = np.empty(n, np.float32)
        t = np.empty(n, np.uint8)
        self._ck(self.lib.dqnhip_read_memory(self.h, first, n, _p(s), _p(a), _p(r), _p(mc), _p(nx),
                                             t.ctypes.data_as(capi.up)))
        return s, a, r, mc, nx, t

    # -- snapshot / restore (src/dqn.cpp:525-620) ---------------------------------------------------
    def Snapshot(self, snapshot_prefix=None, remove_old=None, snapshot_memory=None):
        """DQN::Snapshot(): `<prefix>_{actor,critic}_iter_N.{caffemodel,solverstate}` (+
        `<prefix>_iter_N.replaymemory`) in Caffe's binary protobuf layout.  With no arguments:
        Snapshot(save_path_, FLAGS_remove_old_snapshots=True, FLAGS_snapshot_memory=True)."""
        prefix = self.save_path_ if snapshot_prefix is None else snapshot_prefix
        remove_old = True if remove_old is None and snapshot_prefix is None else bool(remove_old)
        snapshot_memory = True if snapshot_memory is None else bool(snapshot_memory)
        self._ck(self.lib.dqnhip_snapshot(self.h, os.fsencode(self.save_path_), os.fsencode(prefix),
                                          int(remove_old), int(snapshot_memory)))
        self.last_snapshot_iter_ = self.max_iter()

    def RestoreActorSolver(self, actor_solver):
        self._ck(self.lib.dqnhip_solver_restore(self.h, ACTOR, os.fsencode(actor_solver)))
        self.last_snapshot_iter_ = self.max_iter()

    def RestoreCriticSolver(self, critic_solver):
        self._ck(self.lib.dqnhip_solver_restore(self.h, CRITIC, os.fsencode(critic_solver)))
        self.last_snapshot_iter_ = self.max_iter()

    def LoadActorWeights(self, actor_weights):
        self._ck(self.lib.dqnhip_load_caffemodel(self.h, ACTOR, os.fsencode(actor_weights)))

    def LoadCriticWeights(self, critic_weights):
        self._ck(self.lib.dqnhip_load_caffemodel(self.h, CRITIC, os.fsencode(critic_weights)))

    def SnapshotReplayMemory(self, filename):
        """src/dqn.cpp:1146-1178: gzip `.replaymemory` file in the reference's byte layout."""
        self._ck(self.lib.dqnhip_snapshot_replay_memory(self.h, os.fsencode(filename)))

    def LoadReplayMemory(self, filename):
        """src/dqn.cpp:1180-1226."""
        self._ck(self.lib.dqnhip_load_replay_memory(self.h, os.fsencode(filename)))

    # -- acting (src/dqn.cpp:664-711) -------------------------------------------------------
    def GetRandomActorOutput(self):
        g = self.random_engine
        ao = np.empty(10, np.float32)
        ao[0:4] = g.uniform(-1.0, 1.0, 4)
        ao[4] = g.uniform(-100.0, 100.0)
        ao[5] = g.uniform(-180.0, 180.0); ao[6] = g.uniform(-180.0, 180.0); ao[7] = g.uniform(-180.0, 180.0)
        ao[8] = g.uniform(0.0, 100.0)
        ao[9] = g.uniform(-180.0, 180.0)
        return ao

    def SelectActionGreedily(self, states_batch, net=ACTOR):
        s = _f32(states_batch).reshape(-1, self.state_size_)
        out = np.empty((s.shape[0], 10), np.float32)
        self._ck(self.lib.dqnhip_select_actions_net(self.h, net, _p(s), s.shape[0], _p(out)))
        return out

    def SelectActions(self, states_batch, epsilon):
        """src/dqn.cpp:695-711: ONE epsilon draw for the whole batch.  The reference's `CHECK_LE(states_batch.size(),
        kMinibatchSize)` (:699) holds with this learner's minibatch; `select_actions_cap` widens it (-1: any batch — the
        device path has no MemoryData layer whose width would bound it; n > 0: that many)."""
        if not (0.0 <= epsilon <= 1.0):
            raise DQNFatal("CHECK failed: epsilon >= 0.0 && epsilon <= 1.0")
        s = _f32(states_batch).reshape(-1, self.state_size_)
        cap = self.kMinibatchSize if self.select_actions_cap == 0 else self.select_actions_cap
        if cap >= 0 and s.shape[0] > cap:
            raise DQNFatal("CHECK failed: states_batch.size() <= kMinibatchSize (%d vs. %d)" % (s.shape[0], cap))
        if self.random_engine.uniform(0.0, 1.0) < epsilon:
            return np.stack([self.GetRandomActorOutput() for _ in range(s.shape[0])])
        return self.SelectActionGreedily(s)

    def SelectAction(self, input_states, epsilon):
        return self.SelectActions(np.asarray(input_states, np.float32).reshape(1, -1), epsilon)[0]

    def SampleAction(self, actor_output):
        """src/dqn.cpp:180-194: sample the discrete action with probabilities max(0, logit+1),
        TACKLE removed; not used on the reference's default path."""
        ao = np.asarray(actor_output, dtype=np.float32)
        p = np.array([max(0.0, ao[DASH] + 1.0), max(0.0, ao[TURN] + 1.0), 0.0, max(0.0, ao[KICK] + 1.0)])
        act = int(self.random_engine.choice(4, p=p / p.sum()))
        o1, o2 = GetParamOffset(act, 0), GetParamOffset(act, 1)
        return Action(act, float(ao[kActionSize + o1]), 0.0 if o2 < 0 else float(ao[kActionSize + o2]))

    def SampleStatesFromMemory(self, n, idx=None):
        """src/dqn.cpp:511-523: the states of n uniformly sampled transitions ([n,S]); idx = the
        explicit form of SampleTransitionsFromMemory, None = device draw."""
        out = np.empty((n, self.state_size_), np.float32)
        ip = None
        if idx is not None:
            i = np.ascontiguousarray(idx, dtype=np.int32)
            if i.size != n:
                raise DQNFatal("need %d indices" % n)
            ip = i.ctypes.data_as(capi.ip)
        self._ck(self.lib.dqnhip_sample_states(self.h, ip, n, _p(out)))
        return out

    def getActorOutput(self, batch_size, net=ACTOR):
        """src/dqn.cpp:719-732: the actor's output blobs as its last minibatch forward left them."""
        out = np.empty((batch_size, 10), np.float32)
        self._ck(self.lib.dqnhip_get_actor_output(self.h, net, batch_size, _p(out)))
        return out

    def CriticForward(self, states_batch, action_batch, net=CRITIC):
        s = _f32(states_batch).reshape(-1, self.state_size_)
        a = _f32(action_batch).reshape(-1, 10)
        if s.shape[0] != a.shape[0]:
            raise DQNFatal("CHECK_EQ(states_batch.size(), action_batch.size())")
        q = np.empty(s.shape[0], np.float32)
        self._ck(self.lib.dqnhip_critic_forward(self.h, net, _p(s), _p(a), s.shape[0], _p(q)))
        return q

    def EvaluateAction(self, input_states, action):
        """src/dqn.cpp:688-693."""
        return float(self.CriticForward(np.asarray(input_states).reshape(1, -1), np.asarray(action).reshape(1, -1))[0])

    # -- the update (src/dqn.cpp:799-972) ---------------------------------------------------------
    def UpdateActorCritic(self, idx=None):
        """One update; idx = explicit sampled indices or None for on-device
        sampling.  Returns (critic_loss, avg_q) like the reference."""
        loss, avgq = C.c_float(), C.c_float()
        keep, ip = self._idx(idx)
        self._ck(self.lib.dqnhip_update(self.h, ip, C.byref(loss), C.byref(avgq)))
        return loss.value, avgq.value

    def _idx(self, idx):
        """B explicit sampled indices as an int32 pointer (the C side reads exactly kMinibatchSize)."""
        if idx is None:
            return None, None
        i = np.ascontiguousarray(idx, dtype=np.int32)
        if i.size != self.kMinibatchSize:
            raise DQNFatal("need %d indices, got %d" % (self.kMinibatchSize, i.size))
        return i, i.ctypes.data_as(capi.ip)

    def UpdateActorCriticChained(self, idx, idx_next=None):
        """dqnhip_update_chained: this update on idx; idx_next = the indices the NEXT call will bring (its gather and first layers then
        ride in this update's optimiser launches).  Same results as UpdateActorCritic(idx)."""
        loss, avgq = C.c_float(), C.c_float()
        keep, ip = self._idx(idx)
        keep2, ip2 = self._idx(idx_next)
        self._ck(self.lib.dqnhip_update_chained(self.h, ip, ip2, C.byref(loss), C.byref(avgq)))
        return loss.value, avgq.value

    def UpdateActorCriticPipelined(self, idx=None):
        """dqnhip_update_pipelined: enqueue this update, return (critic_loss, avg_q) of the PREVIOUS one."""
        loss, avgq = C.c_float(), C.c_float()
        keep, ip = self._idx(idx)
        self._ck(self.lib.dqnhip_update_pipelined(self.h, ip, C.byref(loss), C.byref(avgq)))
        return loss.value, avgq.value

    def update_async(self, idx=None):
        keep, ip = self._idx(idx)
        self._ck(self.lib.dqnhip_update_async(self.h, ip))

    def update_async_n(self, n):
        """n updates with on-device sampling in one call (dqnhip_update_async_n)."""
        self._ck(self.lib.dqnhip_update_async_n(self.h, int(n)))

    def update_phase(self, phase, idx=None):
        keep, ip = self._idx(idx)
        self._ck(self.lib.dqnhip_update_phase(self.h, phase, ip))

    def update_abort(self):
        """Abandon a phased update whose exchange step failed (the next update starts afresh)."""
        self._ck(self.lib.dqnhip_update_abort(self.h))

    def apply_update(self, net):
        """Solver::ApplyUpdate of one net on the gradient in its arena (src/dqn.cpp:904 tail, :964-965)."""
        self._ck(self.lib.dqnhip_apply_update(self.h, net))

    def apply_update_sharded(self, net, world):
        """The same step, evaluated as a `world`-rank group with a sharded optimiser evaluates it (one slice after the other)."""
        self._ck(self.lib.dqnhip_apply_update_sharded(self.h, net, int(world)))

    # -- native data parallelism (RCCL inside the library, include/dqnhip.h dqnhip_dp_*) -------
    @staticmethod
    def dp_unique_id():
        lib = capi.load()
        buf = C.create_string_buffer(capi.DP_ID_BYTES)
        if lib.dqnhip_dp_unique_id(buf, capi.DP_ID_BYTES) != 0:
            raise DQNFatal(lib.dqnhip_last_error().decode())
        return buf.raw

    @staticmethod
    def _dp_flags(per_layer, half_grads, shard_opt=False, unverified_ok=False):
        return ((capi.DP_PER_LAYER if per_layer else 0) | (capi.DP_HALF_GRADS if half_grads else 0) |
                (capi.DP_SHARD_OPT if shard_opt else 0) | (capi.DP_UNVERIFIED_OK if unverified_ok else 0))

    def dp_init(self, unique_id, per_layer=False, half_grads=False, shard_opt=False, unverified_ok=False):
        """unverified_ok: per_layer / shard_opt have never run on more than one rank; a real group must ask for them explicitly."""
        assert len(unique_id) == capi.DP_ID_BYTES
        buf = C.create_string_buffer(bytes(unique_id), capi.DP_ID_BYTES)
        self._ck(self.lib.dqnhip_dp_init(self.h, buf, capi.DP_ID_BYTES, self._dp_flags(per_layer, half_grads, shard_opt, unverified_ok)))

    def dp_init_file(self, path, per_layer=False, timeout_s=120, half_grads=False, shard_opt=False, unverified_ok=False):
        self._ck(self.lib.dqnhip_dp_init_file(self.h, os.fsencode(path), self._dp_flags(per_layer, half_grads, shard_opt, unverified_ok), int(timeout_s)))

    @staticmethod
    def dp_info():
        """(RCCL version, path of the librccl this process resolved): a group must not mix builds (dqnhip_dp_init checks)."""
        lib = capi.load()
        v = C.c_int32()
        buf = C.create_string_buffer(4096)
        if lib.dqnhip_dp_info(C.byref(v), buf, 4096) != 0:
            raise DQNFatal(lib.dqnhip_last_error().decode())
        return v.value, os.fsdecode(buf.value)

    def dp_gather_state(self):
        """sharded optimiser: all-gather of the Adam history (collective; no-op otherwise)"""
        self._ck(self.lib.dqnhip_dp_gather_state(self.h))

    def dp_graph_active(self):
        v = C.c_int32()
        self._ck(self.lib.dqnhip_dp_graph_active(self.h, C.byref(v)))
        return bool(v.value)

    def dp_destroy(self):
        self._ck(self.lib.dqnhip_dp_destroy(self.h))

    def dp_broadcast_params(self, root=0):
        self._ck(self.lib.dqnhip_dp_broadcast_params(self.h, int(root)))

    def dp_update(self, idx=None):
        keep, ip = self._idx(idx)
        self._ck(self.lib.dqnhip_dp_update(self.h, ip))

    def dp_update_n(self, n):
        """n data-parallel updates with on-device sampling in one call (every rank: the same n)."""
        self._ck(self.lib.dqnhip_dp_update_n(self.h, int(n)))

    def skipped_steps(self):
        n = C.c_int64()
        self._ck(self.lib.dqnhip_skipped_steps(self.h, C.byref(n)))
        return n.value

    def read_stats(self):
        loss, avgq = C.c_float(), C.c_float()
        self._ck(self.lib.dqnhip_read_stats(self.h, C.byref(loss), C.byref(avgq)))
        return loss.value, avgq.value

    def Update(self, loss_display_iter=1000):
        """src/dqn.cpp:799-826 (snapshotting is in snapshot.py)."""
        if self.memory_size() < self.memory_threshold:
            return None
        critic_loss, avg_q = self.UpdateActorCritic()
        logs = []
        if self.critic_iter() % loss_display_iter == 0:
            logs.append("[Agent%d] Critic Iteration %d, loss = %g" % (self.tid_, self.critic_iter(), self.smoothed_critic_loss_))
            self.smoothed_critic_loss_ = 0.0
        self.smoothed_critic_loss_ += critic_loss / float(loss_display_iter)
        if self.actor_iter() % loss_display_iter == 0:
            logs.append("[Agent%d] Actor Iteration %d, avg_q_value = %g" % (self.tid_, self.actor_iter(), self.smoothed_actor_loss_))
            self.smoothed_actor_loss_ = 0.0
        self.smoothed_actor_loss_ += avg_q / float(loss_display_iter)
        for line in logs:
            print(line)
        # periodic snapshot (src/dqn.cpp:818-825)
        if (self.critic_iter() >= self.last_snapshot_iter_ + self.snapshot_freq or
                self.actor_iter() >= self.last_snapshot_iter_ + self.snapshot_freq):
            self.Snapshot()
        return critic_loss, avg_q

    def Benchmark(self, iterations=1000, warmup=0):
        """src/dqn.cpp:487-498; returns the average update time in ms."""
        ms = C.c_float()
        self._ck(self.lib.dqnhip_benchmark(self.h, warmup, iterations, C.byref(ms)))
        return ms.value

    def BenchmarkBlocking(self, iterations=1000, warmup=50, seed=1, pipelined=False):
        """DQN::Benchmark as the drop-in's caller sees it: host-drawn indices + a (loss, avg_q) read-back per
        update (blocking, or one-deep pipelined); average wall-clock ms per update."""
        ms = C.c_float()
        self._ck(self.lib.dqnhip_benchmark_blocking(self.h, warmup, iterations, seed, int(pipelined), C.byref(ms)))
        return ms.value

    # -- parameters ------------------------------------------------------------------------------------
    def param_count(self, net):
        n = C.c_size_t()
        self._ck(self.lib.dqnhip_param_count(self.h, net, C.byref(n)))
        return n.value

    def get_params(self, net, kind=KIND_W):
        out = np.empty(self.param_count(net), np.float32)
        self._ck(self.lib.dqnhip_get_params(self.h, net, kind, _p(out), out.size))
        return out

    def set_params(self, net, arr, kind=KIND_W):
        a = _f32(arr)
        self._ck(self.lib.dqnhip_set_params(self.h, net, kind, _p(a), a.size))

    def CloneNet(self, net):
        """src/dqn.cpp:1022-1035: hard copy online -> target."""
        self._ck(self.lib.dqnhip_clone_to_target(self.h, net))

    def ShareParameters(self, other, num_actor_layers_to_share, num_critic_layers_to_share):
        """src/dqn.cpp:1047-1078: `other`'s first layers (and its targets') use this learner's weights."""
        self._ck(self.lib.dqnhip_share_parameters(self.h, other.h, int(num_actor_layers_to_share),
                                                  int(num_critic_layers_to_share)))
        other._share_keepalive = getattr(other, "_share_keepalive", []) + [self]

    def ShareReplayMemory(self, other):
        """src/dqn.cpp:1080-1082: `other` uses this learner's replay memory."""
        self._ck(self.lib.dqnhip_share_replay_memory(self.h, other.h))
        other._share_keepalive = getattr(other, "_share_keepalive", []) + [self]

    def debug_read(self, name):
        B = self.kMinibatchSize
        if name.startswith("act") and name[3:4].isdigit():      # "act<pass>_<layer>": stored tower activations [B][width]
            width = self.cfg.hidden[int(name.split("_")[1]) - 1]
            out = np.empty(B * width, np.float32)
            self._ck(self.lib.dqnhip_debug_read(self.h, name.encode(), _p(out), out.size))
            return out.reshape(B, width)
        wide = name in ("actor_out", "dq_da")
        out = np.empty(B * (10 if wide else 1), np.float32)
        self._ck(self.lib.dqnhip_debug_read(self.h, name.encode(), _p(out), out.size))
        return out.reshape(B, 10) if wide else out

    def grad_buffer(self, net):
        ptr, n = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.dqnhip_grad_buffer(self.h, net, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def stream(self):
        s = C.c_void_p()
        self._ck(self.lib.dqnhip_get_stream(self.h, C.byref(s)))
        return s.value

    def update_plan(self):
        """The launch plan of this learner's update (dqnhip_get_update_plan): the merged forms it takes, by name, and the kernels per
        update counted from a capture of the sequence dqnhip_update* enqueues."""
        p = capi.UpdatePlan()
        p.struct_size = C.sizeof(capi.UpdatePlan)
        self._ck(self.lib.dqnhip_get_update_plan(self.h, C.byref(p)))
        return {"forms": [n for i, n in enumerate(capi.PLAN_FORMS) if p.forms >> i & 1], "launches_single": p.launches_single,
                "launches_graph_first": p.launches_graph_first, "launches_in_graph": p.launches_in_graph,
                "updates_per_graph": p.updates_per_graph, "collectives": p.collectives}

    def set_kernel_timing(self, enable):
        self._ck(self.lib.dqnhip_set_kernel_timing(self.h, int(enable)))

    def kernel_timing(self, family, reset=False):
        ms, n = C.c_float(), C.c_int64()
        self._ck(self.lib.dqnhip_get_kernel_timing(self.h, family.encode(), C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value


def FindLatestSnapshot(snapshot_prefix):
    """src/dqn.cpp:122-144 -> (actor_solverstate, critic_solverstate, replaymemory), '' if none."""
    lib = capi.load()
    bufs = [C.create_string_buffer(4096) for _ in range(3)]
    if lib.dqnhip_find_latest_snapshot(os.fsencode(snapshot_prefix), bufs[0], bufs[1], bufs[2], 4096) != 0:
        raise DQNFatal(lib.dqnhip_last_error().decode())
    return tuple(os.fsdecode(b.value) for b in bufs)


def FindHiScore(snapshot_prefix):
    """src/dqn.cpp:146-158."""
    lib = capi.load()
    v = C.c_int32()
    if lib.dqnhip_find_hiscore(os.fsencode(snapshot_prefix), C.byref(v)) != 0:
        raise DQNFatal(lib.dqnhip_last_error().decode())
    return v.value


def RemoveFilesMatchingRegexp(regexp):
    """src/dqn.cpp:92-98."""
    lib = capi.load()
    if lib.dqnhip_remove_files_matching_regexp(os.fsencode(regexp)) != 0:
        raise DQNFatal(lib.dqnhip_last_error().decode())


def FilesMatchingRegexp(regexp):
    """src/dqn.hpp:213-216: regular files in the regexp's directory whose name matches its last component."""
    lib = capi.load()
    n = C.c_int32()
    size = 1 << 16
    while True:
        buf = C.create_string_buffer(size)
        if lib.dqnhip_files_matching_regexp(os.fsencode(regexp), buf, size, C.byref(n)) == 0:
            return [os.fsdecode(x) for x in buf.value.split(b"\n") if x]
        msg = lib.dqnhip_last_error().decode()
        if "buffer too small" not in msg or size > (1 << 26):
            raise DQNFatal(msg)
        size *= 4


def RemoveSnapshots(regexp, min_iter):
    """src/dqn.hpp:222-224, src/dqn.cpp:100-109."""
    lib = capi.load()
    if lib.dqnhip_remove_snapshots(os.fsencode(regexp), int(min_iter)) != 0:
        raise DQNFatal(lib.dqnhip_last_error().decode())


def dp_rendezvous_file(path, rank, world, unique_id=None, timeout_s=120):
    """File rendezvous of a data-parallel group (needs no GPU): rank 0 passes the id, the others receive it."""
    lib = capi.load()
    buf = C.create_string_buffer(bytes(unique_id) if unique_id is not None else b"", capi.DP_ID_BYTES)
    if lib.dqnhip_dp_rendezvous_file(os.fsencode(path), int(rank), int(world), int(timeout_s), buf, capi.DP_ID_BYTES) != 0:
        raise DQNFatal(lib.dqnhip_last_error().decode())
    return buf.raw


def dp_rendezvous_cleanup(path, world):
    capi.load().dqnhip_dp_rendezvous_cleanup(os.fsencode(path), int(world))


def reduce_gradients_local(learners, net):
    """Sum the gradient arenas of co-located learners of one data-parallel group (rank order)."""
    lib = capi.load()
    arr = (capi.H * len(learners))(*[l.h for l in learners])
    if lib.dqnhip_reduce_gradients_local(arr, len(learners), net) != 0:
        raise DQNFatal(lib.dqnhip_last_error().decode())


class EnvFrontEnd:
    """N concurrent (synthetic) HFO workers feeding one learner's device-resident replay:
    the learner side of PlayOneEpisode (src/dqn_main.cpp:97-153), batched.  See
    include/dqnhip_env.h."""

    def __init__(self, dqn, workers, max_steps=500, unum=7, p_end=0.01, p_goal=0.3, seed=1):
        self.dqn, self.N = dqn, workers
        self.lib = dqn.lib
        cfg = capi.EnvConfig()
        cfg.struct_size = C.sizeof(capi.EnvConfig)
        cfg.workers, cfg.max_steps, cfg.unum = workers, max_steps, unum
        cfg.p_end, cfg.p_goal, cfg.seed = p_end, p_goal, seed
        self.h = C.c_void_p()
        dqn._ck(self.lib.dqnhip_env_create(dqn.h, C.byref(cfg), C.byref(self.h)))

    def step(self, epsilon, n_steps=1):
        self.dqn._ck(self.lib.dqnhip_env_step(self.h, float(epsilon), int(n_steps)))

    def stats(self):
        a, b, g = C.c_int64(), C.c_int64(), C.c_int64()
        r = C.c_double()
        self.dqn._ck(self.lib.dqnhip_env_stats(self.h, C.byref(a), C.byref(b), C.byref(r), C.byref(g)))
        return a.value, b.value, r.value, g.value

    def debug_read(self, name):
        N, S = self.N, self.dqn.state_size_
        n = N * (S if name == "state" else 10 if name == "actor_out" else 1)
        out = np.empty(n, np.float32)
        self.dqn._ck(self.lib.dqnhip_env_debug_read(self.h, name.encode(), _p(out), n))
        if name == "state":
            return out.reshape(N, S)
        if name == "actor_out":
            return out.reshape(N, 10)
        return out

    def close(self):
        if self.h:
            self.lib.dqnhip_env_destroy(self.h)
            self.h = None
