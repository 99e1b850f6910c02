"""Host-side mirror of the reference interface (no GPU): GetAction / GetParamOffset /
PrintActorOutput / LabelTransitions."""
import ctypes as C

import numpy as np
import pytest

from oracle import c_oracle


def test_get_action_matches_oracle_and_reference_semantics(pkg):
    rng = np.random.default_rng(0)
    ao = rng.uniform(-1, 1, size=(500, 10)).astype(np.float32)
    ao[:50, 2] = 5.0                     # a dominant TACKLE logit must never win (src/dqn.cpp:198)
    ao[50:60, 0] = ao[50:60, 1] = ao[50:60, 3] = 0.25   # ties -> lowest index
    act, a1, a2 = c_oracle.get_action(ao)
    for i, row in enumerate(ao):
        a = pkg.GetAction(row)
        assert (a.action, a.arg1, a.arg2) == (int(act[i]), float(a1[i]), float(a2[i]))
        assert a.action != pkg.TACKLE
    assert all(pkg.GetAction(r).action == pkg.DASH for r in ao[50:60])


def test_get_param_offset_table(pkg):
    # src/dqn.cpp:162-178
    exp = {(pkg.DASH, 0): 0, (pkg.DASH, 1): 1, (pkg.TURN, 0): 2, (pkg.TURN, 1): -1,
           (pkg.TACKLE, 0): 3, (pkg.TACKLE, 1): -1, (pkg.KICK, 0): 4, (pkg.KICK, 1): 5}
    for (a, n), v in exp.items():
        assert pkg.GetParamOffset(a, n) == v
    assert pkg.GetParamOffset(pkg.DASH, 2) == -1 and pkg.GetParamOffset(pkg.DASH, -1) == -1
    with pytest.raises(pkg.DQNFatal):
        pkg.GetParamOffset(7, 0)


def test_print_actor_output_format(pkg):
    # src/dqn.cpp:210-216 (std::to_string prints 6 decimals)
    s = pkg.PrintActorOutput(np.arange(10, dtype=np.float32))
    assert s == ("Dash(4.000000, 5.000000)=0.000000, Turn(6.000000)=1.000000, Tackle(7.000000)=2.000000, "
                 "Kick(8.000000, 9.000000)=3.000000")


def test_label_transitions_c_abi_matches_oracle(pkg):
    lib = pkg.capi.load()
    rng = np.random.default_rng(1)
    for n in (1, 2, 17, 500):
        r = rng.uniform(-1, 5, size=n).astype(np.float32)
        out = np.empty(n, np.float32)
        rc = lib.dqnhip_label_transitions(0.99, r.ctypes.data_as(pkg.capi.fp), n, out.ctypes.data_as(pkg.capi.fp))
        assert rc == 0
        np.testing.assert_array_equal(out, c_oracle.label_transitions(0.99, r))
    out = np.empty(1, np.float32)
    assert lib.dqnhip_label_transitions(0.99, out.ctypes.data_as(pkg.capi.fp), 0, out.ctypes.data_as(pkg.capi.fp)) != 0
    assert b"at least one transition" in lib.dqnhip_last_error()      # CHECK_GT, src/dqn.cpp:784
