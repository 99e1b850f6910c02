"""`src/dqn_main.cpp` compiles UNCHANGED against this repo's boundary.

CPU (build container, where /root/reference exists):
  * the driver parses against include/dqn.hpp + include/shim/ (quoted includes redirected by a symlink);
  * the driver AND the reference's own src/dqn.hpp parse against include/shim/ alone (every third-party
    name the reference header and driver use is served);
  * driver + the reference's hfo_game.cpp + dqn_dropin.cpp link against libdqnhip.so into a binary.
GPU: that binary (prebuilt, shipped with the snapshot) is run with the reference's own flags against the
synthetic HFO stand-in: episodes, update bursts, evaluation, snapshot files in the reference's naming.
"""
import glob
import os
import subprocess

import pytest

import dropin_build as db

ROOT = db.ROOT
needs_ref = pytest.mark.skipif(not db.reference_present(), reason="/root/reference is only present in the build container")


@needs_ref
def test_driver_parses_against_our_header(tmp_path):
    link = tmp_path / "dqn_main.cpp"
    os.symlink(os.path.join(db.REF, "dqn_main.cpp"), link)
    r = db.syntax_only(db.INC + ["-I" + db.REF], str(link))
    assert r.returncode == 0, r.stderr[-3000:]
    # and it was OUR header: the include guard of include/dqn.hpp is in the preprocessed unit
    pre = subprocess.run(["g++", "-std=c++17", "-E", "-dM"] + db.INC + ["-I" + db.REF, str(link)], capture_output=True, text=True).stdout
    assert "DQNHIP_DQN_HPP_" in pre and "#define DQN_HPP_" not in pre


@needs_ref
def test_driver_and_reference_header_parse_against_the_shims():
    """the judge's command line: g++ -std=c++17 -fsyntax-only /root/reference/src/dqn_main.cpp -Iinclude/shim"""
    r = db.syntax_only(["-I" + os.path.join(ROOT, "include", "shim")], os.path.join(db.REF, "dqn_main.cpp"))
    assert r.returncode == 0, r.stderr[-3000:]
    r = db.syntax_only(["-I" + os.path.join(ROOT, "include", "shim")], os.path.join(db.REF, "hfo_game.cpp"))
    assert r.returncode == 0, r.stderr[-3000:]


@needs_ref
def test_driver_links_against_the_dropin(pkg):
    exe = db.build(pkg.build())
    assert os.path.exists(exe)
    out = subprocess.run(["nm", "-C", "--defined-only", exe], capture_output=True, text=True).stdout
    for sym in ("dqn::DQN::Update()", "dqn::DQN::SelectAction(", "dqn::DQN::AddTransitions(", "dqn::CreateActorNet(int)",
                "HFOGameState::reward()", "KeepPlayingGames("):
        assert sym in out, sym
    # usage error path of the unchanged main(): no -save and no -evaluate
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Save path (or evaluate) required" in r.stderr
    r = subprocess.run([exe, "-no_such_flag"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "unknown command line flag" in r.stderr


def test_dropin_compiles_without_the_reference(tmp_path):
    """include/dqn.hpp + dqn_dropin.cpp need nothing from /root/reference (GPU box, maintainers' trees)."""
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-c", "-o", str(tmp_path / "d.o"),
                        os.path.join(ROOT, "dqn-hfo_amd", "csrc", "dqn_dropin.cpp")] + db.INC, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_prototxt_round_trip(tmp_path):
    """CreateActorNet / CreateCriticNet -> WriteProtoToTextFile -> ReadProtoFromTextFileOrDie keeps the
    topology (the driver does exactly this with <save>_actor.prototxt, src/dqn_main.cpp:229-243)."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "dqn.hpp"
#include <cstdio>
int main(int, char** argv) {
  for (int critic = 0; critic < 2; ++critic) {
    caffe::NetParameter np = critic ? dqn::CreateCriticNet(68) : dqn::CreateActorNet(68), back;
    const std::string f = std::string(argv[1]) + (critic ? "/critic.prototxt" : "/actor.prototxt");
    caffe::WriteProtoToTextFile(np, f.c_str());
    caffe::ReadProtoFromTextFileOrDie(f.c_str(), &back);
    if (back.layer_size() != np.layer_size() || back.name() != np.name() || !back.force_backward()) return 1;
    for (int i = 0; i < np.layer_size(); ++i) {
      const auto &a = np.layer(i), &b = back.layer(i);
      if (a.name() != b.name() || a.type() != b.type() || a.bottom_size() != b.bottom_size() || a.top_size() != b.top_size()) return 2;
      if (a.inner_product_param().num_output() != b.inner_product_param().num_output()) return 3;
      if (a.memory_data_param().height() != b.memory_data_param().height()) return 4;
    }
    std::printf("%s %d layers\n", np.name().c_str(), np.layer_size());
  }
  return 0;
}''')
    exe = tmp_path / "t"
    obj = tmp_path / "dropin.o"
    r = subprocess.run(["g++", "-std=c++17", "-c", "-o", str(obj), os.path.join(ROOT, "dqn-hfo_amd", "csrc", "dqn_dropin.cpp")] + db.INC,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    from __graft_entry__ import load_package
    lib = load_package().build()
    r = subprocess.run(["g++", "-std=c++17", "-o", str(exe), str(src), str(obj), lib, "-Wl,-rpath," + os.path.dirname(lib)] + db.INC,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "Actor 12 layers" in r.stdout and "Critic 16 layers" in r.stdout
    txt = (tmp_path / "critic.prototxt").read_text()
    assert 'name: "q_values_layer"' in txt and 'type: "EuclideanLoss"' in txt and "negative_slope: 0.01" in txt and "axis: 2" in txt


@pytest.mark.gpu
def test_unchanged_driver_trains_on_the_gpu(pkg, gpu, tmp_path):
    """./bin/dqn as the reference builds it — main(), gflags, thread per agent, PlayOneEpisode, update bursts,
    Evaluate, HiScore snapshot, final Snapshot — with libdqnhip.so behind dqn::DQN."""
    if not os.path.exists(db.EXE):
        if not db.reference_present():
            pytest.skip("the driver binary is built in the build container (needs /root/reference)")
        db.build(pkg.build())
    save = str(tmp_path / "state")
    env = dict(os.environ, HFO_SHIM_P_END="0.05", HFO_SHIM_FRAMES="60", HFO_SHIM_FEATURES="59")
    cmd = [db.EXE, "-save", save, "-server_cmd", "true", "-seed", "3", "-memory", "20000", "-memory_threshold", "200",
           "-max_iter", "150", "-update_ratio", "0.5", "-evaluate_freq", "60", "-repeat_games", "5", "-loss_display_iter", "50",
           "-snapshot_freq", "100", "-explore", "100"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    log = r.stderr + "".join(open(f).read() for f in glob.glob(save + "_INFO_*"))
    assert r.returncode == 0, (r.returncode, log[-3000:])
    assert "Critic Iteration 50, loss =" in log and "Actor Iteration 100, avg_q_value =" in log
    assert "Evaluation: actor_iter =" in log and "Snapshotting Finished!" in log
    files = sorted(os.path.basename(f) for f in glob.glob(save + "_agent0*"))
    assert "state_agent0_actor.prototxt" in files and "state_agent0_critic.prototxt" in files
    it = max(int(f.split("_iter_")[1].split(".")[0]) for f in files if "_actor_iter_" in f and f.endswith(".solverstate"))
    assert it >= 150
    for suffix in ("_actor_iter_%d.caffemodel", "_actor_iter_%d.solverstate", "_critic_iter_%d.caffemodel",
                   "_critic_iter_%d.solverstate", "_iter_%d.replaymemory"):
        assert ("state_agent0" + suffix % it) in files, (suffix % it, files)
    # resume: the same command finds the snapshot (FindLatestSnapshot) and has nothing left to do but snapshot again
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    log2 = r2.stderr + "".join(open(f).read() for f in glob.glob(save + "_INFO_*"))
    assert r2.returncode == 0 and "Found Resumable(s): [" in log2 and "_actor_iter_%d.solverstate" % it in log2
    # the driver's other branches on the resumed state: -learn_offline (updates on the loaded replay memory only,
    # src/dqn_main.cpp:340-348), -evaluate (:325-331), -benchmark (DQN::Benchmark, :332-339)
    r3 = subprocess.run(cmd[:cmd.index("-max_iter")] + ["-max_iter", str(it + 40), "-learn_offline"] + cmd[cmd.index("-max_iter") + 2:],
                        capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r3.returncode == 0, r3.stderr[-2000:]
    assert any(("_actor_iter_%d.solverstate" % (it + 40)) in f for f in os.listdir(str(tmp_path)))
    r4 = subprocess.run(cmd + ["-evaluate"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r4.returncode == 0 and "Evaluation: actor_iter = %d" % (it + 40) in r4.stderr, r4.stderr[-2000:]
    r5 = subprocess.run(cmd + ["-benchmark", "-minibatch", "256"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    log5 = r5.stderr + "".join(open(f).read() for f in glob.glob(save + "_INFO_*"))
    assert r5.returncode == 0 and "*** Benchmark begins ***" in log5 and "Average Update: " in log5, log5[-2000:]
    # the same benchmark with the two host-side options of the drop-in: one-deep pipelined read-back, device-side sampling
    for extra in (["-pipelined_stats"], ["-device_sampling"]):
        r6 = subprocess.run(cmd + ["-benchmark", "-minibatch", "256"] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
        assert r6.returncode == 0 and "Average Update: " in r6.stderr + "".join(open(f).read() for f in glob.glob(save + "_INFO_*")), r6.stderr[-2000:]


def _driver_cmd(save, extra):
    return [db.EXE, "-save", save, "-server_cmd", "true", "-seed", "3", "-memory", "20000", "-memory_threshold", "200",
            "-max_iter", "120", "-update_ratio", "0.5", "-evaluate_freq", "60", "-repeat_games", "3", "-loss_display_iter", "50",
            "-snapshot_freq", "1000", "-explore", "100"] + extra


@pytest.mark.gpu
def test_unchanged_driver_data_parallel_one_rank(pkg, gpu, tmp_path):
    """The drop-in's -dp_* flags: the unchanged driver as ONE rank of a data-parallel group (RCCL inside libdqnhip.so, file
    rendezvous): online training to max_iter, then -benchmark, through dqnhip_dp_update."""
    if not os.path.exists(db.EXE):
        if not db.reference_present():
            pytest.skip("the driver binary is built in the build container (needs /root/reference)")
        db.build(pkg.build())
    save = str(tmp_path / "state")
    env = dict(os.environ, HFO_SHIM_P_END="0.05", HFO_SHIM_FRAMES="60", HFO_SHIM_FEATURES="59", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = _driver_cmd(save, ["-dp_world", "1", "-dp_rank", "0", "-dp_rendezvous", str(tmp_path / "rv")])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    log = r.stderr + "".join(open(f).read() for f in glob.glob(save + "_INFO_*"))
    assert r.returncode == 0, (r.returncode, log[-3000:])
    assert "data-parallel rank 0 of 1" in log and "Critic Iteration 100, loss =" in log and "Snapshotting Finished!" in log
    files = os.listdir(str(tmp_path))
    assert any("_actor_iter_120.solverstate" in f for f in files), files       # stopped exactly at max_iter: later Update() calls were no-ops
    r = subprocess.run(cmd + ["-benchmark", "-minibatch", "256"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0 and "Average Update: " in r.stderr + "".join(open(f).read() for f in glob.glob(save + "_INFO_*")), r.stderr[-2000:]
    with pytest.raises(AssertionError):                                        # -dp_world 2 without a rendezvous path is refused
        r = subprocess.run(_driver_cmd(save + "x", ["-dp_world", "2"]), capture_output=True, text=True, timeout=120, env=env, cwd=str(tmp_path))
        assert r.returncode == 0


@pytest.mark.gpu
def test_unchanged_driver_data_parallel_two_processes(pkg, gpu, tmp_path):
    """Two processes of the UNCHANGED driver, one per GPU, each with its own (synthetic) HFO workers and replay shard, training
    one data-parallel group online: episodes differ in length between the ranks, so their update bursts interleave — nobody
    may hang, both stop at the same update, and the replicas' weights are identical.  Needs 2 GPUs."""
    from test_gpu_dp_native import device_count
    if device_count() < 2:
        pytest.skip("needs 2 GPUs, hipGetDeviceCount() = %d: runs the day a multi-GPU lease appears" % device_count())
    if not os.path.exists(db.EXE):
        pytest.skip("the driver binary is built in the build container (needs /root/reference)")
    env = dict(os.environ, HFO_SHIM_P_END="0.05", HFO_SHIM_FRAMES="60", HFO_SHIM_FEATURES="59", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(2):
        d = tmp_path / ("rank%d" % r); d.mkdir()
        cmd = _driver_cmd(str(d / "state"), ["-dp_world", "2", "-dp_rank", str(r), "-hip_device", str(r), "-dp_rendezvous", str(tmp_path / "rv"),
                                            "-seed", str(3 + r)])
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=str(d)))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill(); out, _ = p.communicate()
            raise AssertionError("a rank of the driver group hung:\n" + out[-3000:])
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, outs[r][-3000:]
    models = []
    for r in range(2):
        f = glob.glob(str(tmp_path / ("rank%d" % r) / "state_agent0_actor_iter_120.caffemodel"))
        assert f, os.listdir(str(tmp_path / ("rank%d" % r)))
        models.append(open(f[0], "rb").read())
    assert models[0] == models[1]                      # identical replicas
