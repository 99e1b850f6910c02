"""Caffe snapshot layout: `.caffemodel` / `.solverstate` written by the library parse with an
independent python-protobuf codec (tests/caffe_pb.py) and vice versa; DQN::Snapshot naming,
old-snapshot removal and resume (reference src/dqn.cpp:525-620)."""
import os

import numpy as np
import pytest

from caffe_pb import NetParameter, SolverState
from helpers import make_pair
from oracle.torch_ref import layout

pytestmark = pytest.mark.gpu

S, HID = 59, (128, 64, 64, 64)


def names(actor):
    n = ["ip%d_layer" % (i + 1) for i in range(len(HID))]
    return n + (["action_layer", "actionpara_layer"] if actor else ["q_values_layer"])


def dense_from_net(np_msg, actor):
    by = {l.name: l for l in np_msg.layer}
    parts = []
    for name, (n, k) in zip(names(actor), layout(S if actor else S + 10, HID, (4, 6) if actor else (1,))):
        l = by[name]
        assert l.type == "InnerProduct" and len(l.blobs) == 2
        assert list(l.blobs[0].shape.dim) == [n, k] and list(l.blobs[1].shape.dim) == [n]
        parts += [np.array(l.blobs[0].data, np.float32), np.array(l.blobs[1].data, np.float32)]
    return np.concatenate(parts)


def check_full_layer_list(m, actor):
    """Net::ToProto writes EVERY layer of the initialised net (src/dqn.cpp:589-590), not only those with blobs: the
    layers of CreateActorNet / CreateCriticNet (:399-455) in order, the Split that Net::Init inserts under the actor's
    two heads, phase TRAIN everywhere, and each layer's own parameter message."""
    L = len(HID)
    tower = []
    for i in range(1, L + 1):
        tower += [("ip%d_layer" % i, "InnerProduct"), ("ip%d_relu_layer" % i, "ReLU")]
    if actor:
        sp = "ip%d_ip%d_relu_layer_0_split" % (L, L)
        want = [("state_input_layer", "MemoryData"), ("silence", "Silence")] + tower + \
               [(sp, "Split"), ("action_layer", "InnerProduct"), ("actionpara_layer", "InnerProduct")]
    else:
        want = [("state_input_layer", "MemoryData"), ("action_input_layer", "MemoryData"), ("action_params_input_layer", "MemoryData"),
                ("target_input_layer", "MemoryData"), ("silence", "Silence"), ("concat", "Concat")] + tower + \
               [("q_values_layer", "InnerProduct"), ("loss", "EuclideanLoss")]
    assert [(l.name, l.type) for l in m.layer] == want
    by = {l.name: l for l in m.layer}
    assert all(l.HasField("phase") and l.phase == 0 for l in m.layer)
    md = by["state_input_layer"].memory_data_param
    assert (md.batch_size, md.channels, md.height, md.width) == (32, 1, S, 1) and list(by["state_input_layer"].top) == ["states", "dummy1"]
    ip1 = by["ip1_layer"]
    assert ip1.inner_product_param.num_output == HID[0] and ip1.inner_product_param.weight_filler.type == "gaussian"
    assert abs(ip1.inner_product_param.weight_filler.std - 0.01) < 1e-9
    assert list(ip1.bottom) == ["states" if actor else "state_actions"] and list(ip1.top) == ["ip1"]
    r = by["ip2_relu_layer"]
    assert abs(r.relu_param.negative_slope - 0.01) < 1e-9 and list(r.bottom) == list(r.top) == ["ip2"]       # in place
    if actor:
        assert list(by[sp].bottom) == ["ip%d" % L] and list(by[sp].top) == [sp + "_0", sp + "_1"]
        assert list(by["action_layer"].bottom) == [sp + "_0"] and list(by["actionpara_layer"].bottom) == [sp + "_1"]
        assert by["actionpara_layer"].inner_product_param.num_output == 6
    else:
        assert by["concat"].concat_param.axis == 2 and list(by["concat"].bottom) == ["states", "actions", "action_params"]
        t = by["target_input_layer"].memory_data_param
        assert (t.batch_size, t.channels, t.height, t.width) == (32, 1, 1, 1)
        assert list(by["loss"].bottom) == ["q_values", "target"] and list(by["silence"].bottom) == ["dummy1", "dummy2", "dummy3", "dummy4"]


def test_caffemodel_and_solverstate_layout(pkg, gpu, tmp_path):
    dqn, orc, data, rng = make_pair(pkg, B=32, S=S, hidden=HID, save_path=str(tmp_path / "run_agent0"))
    for _ in range(3):
        idx = rng.integers(0, 2048, size=32); dqn.UpdateActorCritic(idx); orc.update(idx)
    dqn.Snapshot()
    pre = str(tmp_path / "run_agent0")
    for net, tag, actor in ((0, "actor", True), (1, "critic", False)):
        cm, ss = pre + "_%s_iter_3.caffemodel" % tag, pre + "_%s_iter_3.solverstate" % tag
        assert os.path.isfile(cm) and os.path.isfile(ss)
        m = NetParameter(); m.ParseFromString(open(cm, "rb").read())
        assert m.name == ("Actor" if actor else "Critic")
        assert [l.name for l in m.layer if l.blobs] == names(actor)
        check_full_layer_list(m, actor)
        np.testing.assert_array_equal(dense_from_net(m, actor), dqn.get_params(net))
        st = SolverState(); st.ParseFromString(open(ss, "rb").read())
        assert st.iter == 3 and st.current_step == 0 and st.learned_net.endswith("_%s_iter_3.caffemodel" % tag)
        shapes = layout(S if actor else S + 10, HID, (4, 6) if actor else (1,))
        assert len(st.history) == 4 * len(shapes)                      # (W,b) x (m, v)
        hist = np.concatenate([np.array(b.data, np.float32) for b in st.history])
        P = dqn.param_count(net)
        np.testing.assert_array_equal(hist[:P], dqn.get_params(net, 1))  # all m first ...
        np.testing.assert_array_equal(hist[P:], dqn.get_params(net, 2))  # ... then all v
    assert os.path.isfile(pre + "_iter_3.replaymemory")
    assert pkg.FindLatestSnapshot(pre) == (pre + "_actor_iter_3.solverstate", pre + "_critic_iter_3.solverstate",
                                           pre + "_iter_3.replaymemory")
    dqn.close(); orc.close()


def test_resume_equals_reference_semantics(pkg, gpu, tmp_path):
    """Snapshot at iter 3, restore into a fresh learner, continue: identical to a run whose target
    nets were re-cloned at that point — the reference does NOT checkpoint the targets
    (src/dqn.cpp:546,555)."""
    pre = str(tmp_path / "r_agent0")
    dqn, orc, data, rng = make_pair(pkg, B=32, S=S, hidden=HID, save_path=pre)
    idxs = rng.integers(0, 2048, size=(6, 32))
    for u in range(3):
        dqn.UpdateActorCritic(idxs[u]); orc.update(idxs[u])
    dqn.Snapshot()
    a, c, m = pkg.FindLatestSnapshot(pre)
    fresh = pkg.DQN(S, minibatch=32, hidden=HID, memory=4096, seed=99, save_path=pre)
    fresh.RestoreActorSolver(a); fresh.RestoreCriticSolver(c); fresh.LoadReplayMemory(m)
    assert (fresh.actor_iter(), fresh.critic_iter(), fresh.memory_size()) == (3, 3, dqn.memory_size())
    orc.clone_to_target(0); orc.clone_to_target(1)                    # what Restore does to the targets
    for net in range(4):
        np.testing.assert_array_equal(fresh.get_params(net), dqn.get_params(net % 2))   # targets == online weights
    for u in range(3, 6):
        l1, q1 = fresh.UpdateActorCritic(idxs[u]); l2, q2 = orc.update(idxs[u])
        assert abs(l1 - l2) <= 1e-4 * max(1, abs(l2)) and abs(q1 - q2) <= 1e-4
    assert fresh.actor_iter() == 6
    dqn.close(); orc.close(); fresh.close()


def test_remove_old_and_hiscore_snapshots(pkg, gpu, tmp_path):
    pre = str(tmp_path / "g_agent0")
    dqn, orc, data, rng = make_pair(pkg, B=32, S=S, hidden=HID, save_path=pre)
    for u in range(2):
        dqn.UpdateActorCritic(rng.integers(0, 2048, size=32))
    dqn.Snapshot()                                                     # iter 2
    for u in range(2):
        dqn.UpdateActorCritic(rng.integers(0, 2048, size=32))
    dqn.Snapshot()                                                     # iter 4: removes iter < 3
    files = sorted(os.listdir(tmp_path))
    assert all("_iter_4." in f for f in files) and len(files) == 5, files
    hs = pre + "_HiScore0.800000"                                      # src/dqn_main.cpp:372-373
    dqn.Snapshot(hs, False, False)
    assert os.path.isfile(hs + "_actor_iter_4.caffemodel") and not os.path.exists(hs + "_iter_4.replaymemory")
    assert os.path.isfile(pre + "_actor_iter_4.caffemodel") is False   # renamed away, as in the reference (:596)
    dqn.close(); orc.close()


def test_load_weights_from_python_written_caffemodel(pkg, gpu, tmp_path):
    """A file produced by the independent codec (extra layers and fields included, as in a real
    Caffe snapshot) loads by layer NAME; the target net is re-cloned."""
    dqn = pkg.DQN(S, minibatch=32, hidden=HID, memory=100)
    rng = np.random.default_rng(0)
    m = NetParameter(); m.name = "Actor"; m.force_backward = True
    extra = m.layer.add(); extra.name = "state_input_layer"; extra.type = "MemoryData"; extra.top.append("states")
    want = []
    for name, (n, k) in zip(names(True), layout(S, HID, (4, 6))):
        l = m.layer.add(); l.name = name; l.type = "InnerProduct"; l.bottom.append("x"); l.top.append("y")
        w = rng.standard_normal(n * k).astype(np.float32); b = rng.standard_normal(n).astype(np.float32)
        bw = l.blobs.add(); bw.shape.dim.extend([n, k]); bw.data.extend(w.tolist())
        bb = l.blobs.add(); bb.shape.dim.extend([n]); bb.data.extend(b.tolist())
        want += [w, b]
        if name == "ip2_layer":
            relu = m.layer.add(); relu.name = "ip2_relu_layer"; relu.type = "ReLU"
    path = str(tmp_path / "py_actor.caffemodel")
    open(path, "wb").write(m.SerializeToString())
    dqn.LoadActorWeights(path)
    np.testing.assert_array_equal(dqn.get_params(0), np.concatenate(want))
    np.testing.assert_array_equal(dqn.get_params(2), np.concatenate(want))      # CloneNet
    with pytest.raises(pkg.DQNFatal, match="Invalid file"):
        dqn.LoadCriticWeights(str(tmp_path / "nope.caffemodel"))
    with pytest.raises(pkg.DQNFatal):
        dqn.LoadCriticWeights(path)                                              # shape mismatch / no critic layer
    dqn.close()
