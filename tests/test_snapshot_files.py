"""Snapshot file search helpers (pure host code, no GPU): FindLatestSnapshot, FindHiScore,
RemoveFilesMatchingRegexp with the reference's naming (src/dqn.cpp:80-158)."""
import os

import pytest


def touch(p):
    open(p, "wb").close()


def test_find_latest_snapshot(pkg, tmp_path):
    pre = str(tmp_path / "state_agent0")
    assert pkg.FindLatestSnapshot(pre) == ("", "", "")
    for it in (10000, 20000, 9000):
        touch(pre + "_actor_iter_%d.solverstate" % it); touch(pre + "_actor_iter_%d.caffemodel" % it)
    for it in (10000, 19999):
        touch(pre + "_critic_iter_%d.solverstate" % it)
    touch(pre + "_iter_20000.replaymemory"); touch(pre + "_iter_100.replaymemory")
    touch(str(tmp_path / "other_agent0_actor_iter_99999.solverstate"))      # different prefix: ignored
    a, c, m = pkg.FindLatestSnapshot(pre)
    assert a == pre + "_actor_iter_20000.solverstate"
    assert c == pre + "_critic_iter_19999.solverstate"
    assert m == pre + "_iter_20000.replaymemory"


def test_find_hiscore_and_remove(pkg, tmp_path):
    pre = str(tmp_path / "run_agent1")
    assert pkg.FindHiScore(pre) == -2147483648                               # numeric_limits<int>::lowest()
    for score, it in ((3, 100), (17, 200), (-5, 300)):
        touch(pre + "_HiScore%d_iter_%d.caffemodel" % (score, it))
    assert pkg.FindHiScore(pre) == 17
    touch(pre + "_actor_iter_5.caffemodel")
    pkg.RemoveFilesMatchingRegexp(pre + "_HiScore.*")                        # src/dqn_main.cpp:371
    left = sorted(os.listdir(tmp_path))
    assert left == ["run_agent1_actor_iter_5.caffemodel"]


def test_files_matching_regexp_and_remove_snapshots(pkg, tmp_path):
    names = ["dqn_agent0_actor_iter_100.caffemodel", "dqn_agent0_actor_iter_2000.caffemodel",
             "dqn_agent0_actor_iter_30000.solverstate", "dqn_agent0_critic_iter_100.caffemodel", "other.txt"]
    for nme in names:
        (tmp_path / nme).write_bytes(b"x")
    (tmp_path / "dqn_agent0_actor_iter_7.caffemodel").mkdir()            # directories never match (is_regular_file)
    rx = str(tmp_path / r"dqn_agent0_actor_iter_[0-9]+\.caffemodel")
    got = pkg.FilesMatchingRegexp(rx)
    assert [os.path.basename(g) for g in got] == ["dqn_agent0_actor_iter_100.caffemodel", "dqn_agent0_actor_iter_2000.caffemodel"]
    assert pkg.FilesMatchingRegexp(str(tmp_path / "nothing.*")) == []
    assert pkg.FilesMatchingRegexp(str(tmp_path / "missing_dir" / ".*")) == []
    pkg.RemoveSnapshots(rx, 2000)                                         # iter < min_iter goes (src/dqn.cpp:100-109)
    assert [os.path.basename(g) for g in pkg.FilesMatchingRegexp(rx)] == ["dqn_agent0_actor_iter_2000.caffemodel"]
    assert (tmp_path / "dqn_agent0_critic_iter_100.caffemodel").exists() and (tmp_path / "other.txt").exists()
    with pytest.raises(pkg.DQNFatal):
        pkg.FilesMatchingRegexp(str(tmp_path / "bad[regexp"))
