"""ctypes binding of oracle/dqn_oracle.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (dqn-hfo_amd/) never does.  PARITY UNPINNED: see
the header of dqn_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdqn_oracle.so")

MAXL = 8
NOUT = 10

ACTOR, CRITIC, ACTOR_TARGET, CRITIC_TARGET = 0, 1, 2, 3
KIND_W, KIND_M, KIND_V, KIND_G = 0, 1, 2, 3


class OrcConfig(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("S", C.c_int32), ("L", C.c_int32),
        ("hidden", C.c_int32 * MAXL),
        ("capacity", C.c_int32), ("soft_update_freq", C.c_int32),
        ("global_B", C.c_int32), ("mirror_waste", C.c_int32),
        ("gamma", C.c_double), ("beta", C.c_double),
        ("tau", C.c_float),
        ("lr_actor", C.c_float), ("lr_critic", C.c_float),
        ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("clip", C.c_float),
    ]


class OrcGame(C.Structure):
    _fields_ = [
        ("old_ball_prox", C.c_float), ("ball_prox_delta", C.c_float),
        ("old_kickable", C.c_float), ("kickable_delta", C.c_float),
        ("old_ball_dist_goal", C.c_float), ("ball_dist_goal_delta", C.c_float),
        ("steps", C.c_int32), ("episode_over", C.c_int32),
        ("got_kickable_reward", C.c_int32), ("pass_active", C.c_int32),
        ("player_on_ball_unum", C.c_int32), ("old_player_on_ball_unum", C.c_int32),
        ("our_unum", C.c_int32), ("status", C.c_int32),
        ("total_reward", C.c_double), ("extrinsic_reward", C.c_double),
    ]


def build(force=False):
    src = os.path.join(_HERE, "dqn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        up = C.POINTER(C.c_uint8)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcConfig)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_param_count.restype = C.c_size_t
        L.orc_param_count.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_params.argtypes = [C.c_void_p, C.c_int, C.c_int, fp]
        L.orc_set_params.argtypes = [C.c_void_p, C.c_int, C.c_int, fp]
        L.orc_clone_to_target.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_iters.argtypes = [C.c_void_p, ip, ip]
        L.orc_set_iters.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_grad_ptr.restype = fp
        L.orc_grad_ptr.argtypes = [C.c_void_p, C.c_int]
        L.orc_tail_ptr.restype = fp
        L.orc_tail_ptr.argtypes = [C.c_void_p, C.c_int]
        L.orc_add_transition.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, fp, C.c_uint8]
        L.orc_add_transitions.argtypes = [C.c_void_p, fp, fp, fp, fp, fp, up, C.c_int]
        L.orc_memory_size.argtypes = [C.c_void_p]
        L.orc_clear_memory.argtypes = [C.c_void_p]
        L.orc_read_memory.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, fp, fp, fp, fp, up]
        L.orc_label_transitions.argtypes = [C.c_double, fp, C.c_int, fp]
        L.orc_actor_forward.argtypes = [C.c_void_p, C.c_int, fp, C.c_int, fp]
        L.orc_critic_forward.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_int, fp]
        L.orc_get_action.argtypes = [fp, C.c_int, ip, fp, fp]
        L.orc_update_phase.argtypes = [C.c_void_p, C.c_int, ip]
        L.orc_update.argtypes = [C.c_void_p, ip, fp, fp]
        L.orc_apply_update.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_stats_from_tails.argtypes = [C.c_void_p]
        L.orc_last_stats.argtypes = [C.c_void_p, fp, fp]
        L.orc_debug_read.argtypes = [C.c_void_p, C.c_char_p, fp, C.c_size_t]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_philox_index.restype = C.c_int32
        L.orc_philox_index.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int32]
        L.orc_env_create.restype = C.c_void_p
        L.orc_env_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64]
        L.orc_env_destroy.argtypes = [C.c_void_p]
        L.orc_env_step.argtypes = [C.c_void_p, C.c_float, C.c_int]
        L.orc_env_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.orc_env_read.argtypes = [C.c_void_p, ip, fp, fp, fp, fp, ip]
        L.orc_game_update.argtypes = [C.POINTER(OrcGame), fp, C.c_int, C.c_int]
        L.orc_game_reward.restype = C.c_float
        L.orc_game_reward.argtypes = [C.POINTER(OrcGame)]
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_config(B=32, S=59, hidden=(1024, 512, 256, 128), capacity=500000, gamma=0.99,
                beta=0.5, tau=0.001, soft_update_freq=1, lr_actor=1e-5, lr_critic=1e-3,
                beta1=0.95, beta2=0.999, eps=1e-8, clip=10.0, global_B=0, mirror_waste=0):
    """Defaults are the reference's (src/dqn.cpp:21-31, src/dqn_main.cpp:30-37)."""
    c = OrcConfig()
    c.B, c.S, c.L = B, S, len(hidden)
    for i, h in enumerate(hidden):
        c.hidden[i] = h
    c.capacity, c.soft_update_freq, c.global_B, c.mirror_waste = capacity, soft_update_freq, global_B, mirror_waste
    c.gamma, c.beta, c.tau = gamma, beta, tau
    c.lr_actor, c.lr_critic, c.beta1, c.beta2, c.eps, c.clip = lr_actor, lr_critic, beta1, beta2, eps, clip
    return c


class Oracle:
    """CPU restatement of dqn::DQN's hot path (see dqn_oracle.c)."""

    def __init__(self, **kw):
        self.cfg = make_config(**kw)
        self.L = lib()
        self.h = self.L.orc_create(C.byref(self.cfg))
        self.B, self.S = self.cfg.B, self.cfg.S
        self.hidden = tuple(self.cfg.hidden[i] for i in range(self.cfg.L))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # parameters -----------------------------------------------------------
    def param_count(self, net):
        return self.L.orc_param_count(self.h, net)

    def get_params(self, net, kind=KIND_W):
        out = np.empty(self.param_count(net), np.float32)
        assert self.L.orc_get_params(self.h, net, kind, _fp(out)) == 0
        return out

    def set_params(self, net, arr, kind=KIND_W):
        a, p = _f(arr)
        assert a.size == self.param_count(net)
        assert self.L.orc_set_params(self.h, net, kind, p) == 0

    def clone_to_target(self, net):
        self.L.orc_clone_to_target(self.h, net)

    def get_iters(self):
        a, c = C.c_int32(), C.c_int32()
        self.L.orc_get_iters(self.h, C.byref(a), C.byref(c))
        return a.value, c.value

    def set_iters(self, a, c):
        self.L.orc_set_iters(self.h, a, c)

    def grad_view(self, net):
        n = self.param_count(net)
        return np.ctypeslib.as_array(self.L.orc_grad_ptr(self.h, net), shape=(n,))

    def tail_view(self, net):
        return np.ctypeslib.as_array(self.L.orc_tail_ptr(self.h, net), shape=(4,))

    # replay ----------------------------------------------------------------
    def add_transitions(self, s, a, r, mc, nx, term):
        s, ps = _f(s); a, pa = _f(a); r, pr = _f(r); mc, pm = _f(mc); nx, pn = _f(nx)
        t = np.ascontiguousarray(term, dtype=np.uint8)
        n = r.size
        rc = self.L.orc_add_transitions(self.h, ps, pa, pr, pm, pn,
                                        t.ctypes.data_as(C.POINTER(C.c_uint8)), n)
        assert rc == 0, rc

    def add_transition(self, s, a, r, mc, nx, term):
        s, ps = _f(s); a, pa = _f(a)
        nx, pn = _f(nx if nx is not None else np.zeros(self.S, np.float32))
        self.L.orc_add_transition(self.h, ps, pa, float(r), float(mc), pn, int(bool(term)))

    def memory_size(self):
        return self.L.orc_memory_size(self.h)

    def clear_memory(self):
        self.L.orc_clear_memory(self.h)

    def read_memory(self, first, n):
        S = self.S
        s = np.empty((n, S), np.float32); nx = np.empty((n, S), np.float32)
        a = np.empty((n, NOUT), np.float32); r = np.empty(n, np.float32); mc = np.empty(n, np.float32)
        t = np.empty(n, np.uint8)
        self.L.orc_read_memory(self.h, first, n, _fp(s), _fp(a), _fp(r), _fp(mc), _fp(nx),
                               t.ctypes.data_as(C.POINTER(C.c_uint8)))
        return s, a, r, mc, nx, t

    # acting ----------------------------------------------------------------
    def actor_forward(self, states, net=ACTOR):
        s, ps = _f(states)
        n = s.shape[0]
        out = np.empty((n, NOUT), np.float32)
        self.L.orc_actor_forward(self.h, net, ps, n, _fp(out))
        return out

    def critic_forward(self, states, actor_out, net=CRITIC):
        s, ps = _f(states); a, pa = _f(actor_out)
        n = s.shape[0]
        q = np.empty(n, np.float32)
        self.L.orc_critic_forward(self.h, net, ps, pa, n, _fp(q))
        return q

    # update ----------------------------------------------------------------
    def update(self, idx):
        i = np.ascontiguousarray(idx, dtype=np.int32)
        assert i.size == self.B
        loss, avgq = C.c_float(), C.c_float()
        rc = self.L.orc_update(self.h, i.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(loss), C.byref(avgq))
        assert rc == 0, rc
        return loss.value, avgq.value

    def update_phase(self, phase, idx):
        i = np.ascontiguousarray(idx, dtype=np.int32)
        rc = self.L.orc_update_phase(self.h, phase, i.ctypes.data_as(C.POINTER(C.c_int32)))
        assert rc == 0, rc

    def apply_update(self, net):
        """One solver's ApplyUpdate on the gradient in grad_view(net) (+ that net's soft update, ++iter)."""
        rc = self.L.orc_apply_update(self.h, net)
        assert rc == 0, rc

    def set_stats_from_tails(self):
        self.L.orc_set_stats_from_tails(self.h)

    def last_stats(self):
        loss, avgq = C.c_float(), C.c_float()
        self.L.orc_last_stats(self.h, C.byref(loss), C.byref(avgq))
        return loss.value, avgq.value

    def debug_read(self, name):
        if name.startswith("actA_") or name.startswith("actC_"):      # stored tower activations of the last forward, [B][width]
            width = self.hidden[int(name[5:]) - 1]
            out = np.empty(self.B * width, np.float32)
            rc = self.L.orc_debug_read(self.h, name.encode(), _fp(out), out.size)
            assert rc == 0, (name, rc)
            return out.reshape(self.B, width)
        n = self.B * (NOUT if name in ("actor_out", "dq_da") else 1)
        out = np.empty(n, np.float32)
        rc = self.L.orc_debug_read(self.h, name.encode(), _fp(out), n)
        assert rc == 0, (name, rc)
        return out.reshape(self.B, -1) if name in ("actor_out", "dq_da") else out


def set_threads(n):
    lib().orc_set_threads(int(n))


def usable_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def label_transitions(gamma, rewards):
    r, pr = _f(rewards)
    out = np.empty_like(r)
    lib().orc_label_transitions(float(gamma), pr, r.size, _fp(out))
    return out


def get_action(actor_out):
    a, pa = _f(np.atleast_2d(actor_out))
    n = a.shape[0]
    act = np.empty(n, np.int32); a1 = np.empty(n, np.float32); a2 = np.empty(n, np.float32)
    lib().orc_get_action(pa, n, act.ctypes.data_as(C.POINTER(C.c_int32)), _fp(a1), _fp(a2))
    return act, a1, a2


class GameState:
    """HFOGameState (src/hfo_game.cpp:109-236) without the HFO I/O."""

    def __init__(self, unum=0):
        self.g = OrcGame()
        self.g.our_unum = unum

    def update(self, state, status=0, player_on_ball=0):
        s, ps = _f(state)
        lib().orc_game_update(C.byref(self.g), ps, int(status), int(player_on_ball))

    def reward(self):
        return lib().orc_game_reward(C.byref(self.g))


def philox_indices(seed, update_counter, B, size):
    """What the learner's on-device sampler draws for update `update_counter`."""
    L = lib()
    return np.array([L.orc_philox_index(seed, update_counter, r, size) for r in range(B)], np.int32)


class OracleEnv:
    """N-worker env front-end on the CPU oracle (same synthetic state stream as the device)."""

    def __init__(self, orc, workers, max_steps=500, unum=7, p_end=0.01, p_goal=0.3, seed=1):
        self.orc, self.N = orc, workers
        self.L = lib()
        self.h = self.L.orc_env_create(orc.h, workers, max_steps, unum, p_end, p_goal, seed)

    def step(self, epsilon, n_steps=1):
        self.L.orc_env_step(self.h, float(epsilon), int(n_steps))

    def stats(self):
        a, b, g = C.c_int64(), C.c_int64(), C.c_int64()
        r = C.c_double()
        self.L.orc_env_stats(self.h, C.byref(a), C.byref(b), C.byref(r), C.byref(g))
        return a.value, b.value, r.value, g.value

    def read(self):
        N, S = self.N, self.orc.S
        act = np.empty(N, np.int32); ln = np.empty(N, np.int32)
        a1 = np.empty(N, np.float32); a2 = np.empty(N, np.float32); rw = np.empty(N, np.float32)
        st = np.empty((N, S), np.float32)
        ip = C.POINTER(C.c_int32)
        self.L.orc_env_read(self.h, act.ctypes.data_as(ip), _fp(a1), _fp(a2), _fp(rw), _fp(st), ln.ctypes.data_as(ip))
        return dict(action=act, arg1=a1, arg2=a2, reward=rw, state=st, episode_len=ln)

    def close(self):
        if self.h:
            self.L.orc_env_destroy(self.h)
            self.h = None
