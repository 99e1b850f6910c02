"""`.replaymemory` files (reference src/dqn.cpp:1146-1226): byte layout checked against an
independent Python gzip/struct codec written from the reference's write/read loops."""
import gzip
import struct

import numpy as np
import pytest

from synth import synth_replay

pytestmark = pytest.mark.gpu


def py_write(path, S, s, a, r, mc, term):
    """SnapshotReplayMemory (src/dqn.cpp:1146-1178), kStateInputCount == 1."""
    with gzip.open(path, "wb") as f:
        f.write(struct.pack("<i", len(r)))
        for i in range(len(r)):
            f.write(s[i].astype("<f4").tobytes())          # curr_state
            f.write(a[i].astype("<f4").tobytes())          # sizeof(ActorOutput) = 40
            f.write(struct.pack("<f", r[i]))
            f.write(struct.pack("<f", mc[i]))
            f.write(struct.pack("<?", bool(term[i])))


def py_read(path, S):
    """LoadReplayMemory (src/dqn.cpp:1180-1226): next state of i-1 is state i when i-1 was not
    terminal; a trailing non-terminal stays boost::none."""
    with gzip.open(path, "rb") as f:
        (n,) = struct.unpack("<i", f.read(4))
        s = np.empty((n, S), np.float32); a = np.empty((n, 10), np.float32)
        r = np.empty(n, np.float32); mc = np.empty(n, np.float32); t = np.empty(n, np.uint8)
        for i in range(n):
            s[i] = np.frombuffer(f.read(4 * S), "<f4"); a[i] = np.frombuffer(f.read(40), "<f4")
            r[i], mc[i] = struct.unpack("<ff", f.read(8))
            (t[i],) = struct.unpack("<?", f.read(1))
        assert f.read(1) == b""
    nx = np.zeros((n, S), np.float32)
    none = np.ones(n, np.uint8)
    for i in range(1, n):
        if not t[i - 1]:
            nx[i - 1] = s[i]; none[i - 1] = 0
    return s, a, r, mc, nx, none


@pytest.mark.parametrize("S,n", [(59, 1000), (68, 70000)])
def test_snapshot_layout_and_roundtrip(pkg, gpu, tmp_path, S, n):
    rng = np.random.default_rng(S)
    data = synth_replay(rng, n, S, mean_len=25)
    dqn = pkg.DQN(S, minibatch=32, hidden=(64,), memory=n + 10)
    dqn.add_transitions_arrays(*data)
    path = str(tmp_path / "agent0_iter_7.replaymemory")
    dqn.SnapshotReplayMemory(path)
    got = py_read(path, S)
    for x, y in zip(got, data):
        np.testing.assert_array_equal(x, y)
    # an independently written file loads to the same memory
    path2 = str(tmp_path / "py.replaymemory")
    py_write(path2, S, *[data[k] for k in (0, 1, 2, 3, 5)])
    dqn2 = pkg.DQN(S, minibatch=32, hidden=(64,), memory=n + 10)
    dqn2.add_transitions_arrays(*synth_replay(rng, 50, S))       # LoadReplayMemory clears first
    dqn2.LoadReplayMemory(path2)
    assert dqn2.memory_size() == n
    for x, y in zip(dqn2.read_memory(0, n), data):
        np.testing.assert_array_equal(x, y)
    dqn.close(); dqn2.close()


def test_trailing_non_terminal_loads_as_terminal(pkg, gpu, tmp_path):
    S = 59
    rng = np.random.default_rng(1)
    s, a, r, mc, nx, term = synth_replay(rng, 40, S, mean_len=8)
    term[-1] = 0                                         # cut the file in the middle of an episode
    path = str(tmp_path / "cut.replaymemory")
    py_write(path, S, s, a, r, mc, term)
    dqn = pkg.DQN(S, minibatch=32, hidden=(64,), memory=100)
    dqn.LoadReplayMemory(path)
    got = dqn.read_memory(0, 40)
    assert got[5][-1] == 1 and not got[4][-1].any()      # next_state = boost::none
    np.testing.assert_array_equal(got[5][:-1], term[:-1])
    dqn.close()


def test_load_errors(pkg, gpu, tmp_path):
    dqn = pkg.DQN(59, minibatch=32, hidden=(64,), memory=100)
    with pytest.raises(pkg.DQNFatal, match="Invalid file"):
        dqn.LoadReplayMemory(str(tmp_path / "missing.replaymemory"))
    rng = np.random.default_rng(2)
    s, a, r, mc, nx, term = synth_replay(rng, 200, 59)
    path = str(tmp_path / "big.replaymemory")
    py_write(path, 59, s, a, r, mc, term)
    with pytest.raises(pkg.DQNFatal, match="capacity"):
        dqn.LoadReplayMemory(path)
    dqn.close()


def test_snapshot_of_a_wrapped_ring(pkg, gpu, tmp_path):
    """After eviction the deque's front is in the middle of the ring: the file holds the logical order."""
    S, cap = 59, 300
    rng = np.random.default_rng(4)
    dqn = pkg.DQN(S, minibatch=32, hidden=(64,), memory=cap)
    chunks = [synth_replay(rng, n, S, mean_len=9) for n in (200, 150, 120, 90)]
    for c in chunks:
        dqn.add_transitions_arrays(*c)
    assert dqn.memory_size() == cap - 1
    allrows = [np.concatenate([c[k] for c in chunks]) for k in range(6)]
    want = [x[-(cap - 1):] for x in allrows]                     # the last cap-1 transitions survive
    path = str(tmp_path / "wrapped.replaymemory")
    dqn.SnapshotReplayMemory(path)
    got = py_read(path, S)
    for k in (0, 1, 2, 3):
        np.testing.assert_array_equal(got[k], want[k])
    dqn2 = pkg.DQN(S, minibatch=32, hidden=(64,), memory=cap)
    dqn2.LoadReplayMemory(path)
    assert dqn2.memory_size() == cap - 1
    a, b = dqn2.read_memory(0, cap - 1), dqn.read_memory(0, cap - 1)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    dqn.close(); dqn2.close()
