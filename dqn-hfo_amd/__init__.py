"""dqn-hfo on MI355X: a from-scratch gfx950 implementation of the reference's
`DQN::Update` hot path (replay gather -> actor/critic MLP fwd/bwd -> TD target ->
clip + Adam -> soft target update) behind the reference's own `dqn::DQN` method
surface.  The compute lives in csrc/ (hand-written HIP, C-ABI in include/dqnhip.h);
this package is the thin host-side mirror of the reference interface.

Import name: `dqn_hfo_amd` (the directory is `dqn-hfo_amd/`, which Python cannot
import by name; see load_package() in __graft_entry__.py).
"""
from . import capi
from ._build import build
from .learner import (DQN, DQNFatal, EnvFrontEnd, FindLatestSnapshot, FindHiScore, RemoveFilesMatchingRegexp, FilesMatchingRegexp, RemoveSnapshots, reduce_gradients_local, dp_rendezvous_file, dp_rendezvous_cleanup, Action, Transition, GetAction, GetParamOffset, PrintActorOutput,
                      ACTOR, CRITIC, ACTOR_TARGET, CRITIC_TARGET, KIND_W, KIND_M, KIND_V, KIND_G,
                      DASH, TURN, TACKLE, KICK)

__all__ = ["capi", "build", "DQN", "DQNFatal", "EnvFrontEnd", "FindLatestSnapshot", "FindHiScore", "RemoveFilesMatchingRegexp", "FilesMatchingRegexp", "RemoveSnapshots", "reduce_gradients_local", "dp_rendezvous_file", "dp_rendezvous_cleanup", "Action", "Transition", "GetAction", "GetParamOffset",
           "PrintActorOutput", "ACTOR", "CRITIC", "ACTOR_TARGET", "CRITIC_TARGET", "KIND_W", "KIND_M",
           "KIND_V", "KIND_G", "DASH", "TURN", "TACKLE", "KICK"]
