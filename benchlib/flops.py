"""Algorithmic work of one update and the peaks it is priced against: what `bench.py` divides by."""


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA dense peak


def tower_weights(in_dim, hidden):
    dims = (in_dim,) + tuple(hidden)
    return [dims[i] * dims[i + 1] for i in range(len(hidden))]


def family_flops(B, S, hidden, shifted=True):
    """Algorithmic GEMM FLOPs per update, per kernel (tower layers only; the skinny heads are
    separate kernels).  See DESIGN.md §5.  Keys are the learner's timing families.
    shifted: the backward schedule of learner.hip tower_backward (default for >= 2 tower layers at minibatches whose layers
    take the one-workgroup-type form): dgrad(L-1) | wgrad(i+1) + dgrad(i) ... | wgrad(1) + wgrad(0); else wgrad(i) + dgrad(i)
    per layer and the first layer's wgrad alone (DQNHIP_TUNE_BWD_UNSHIFTED, or the side-by-side pair launches of small shapes)."""
    wa = tower_weights(S, hidden)
    wc = tower_weights(S + 10, hidden)
    h1 = hidden[0]
    L = len(hidden)
    if shifted and L >= 2:
        pair = lambda w: sum(w[j + 1] + w[j] for j in range(1, L - 1))
        return {
            "gemm_fwd_lds_4x2": 2 * B * (2 * sum(wa[1:]) + 2 * sum(wc[1:])),
            "gemm_fwd_lds_2x2": 2 * B * sum(wc[1:]),
            "gemm_fwd_direct": 2 * B * (2 * wa[0] + 3 * wc[0]),
            # wgrad(i+1) + dgrad(i), i = L-2 .. 1: critic train + actor
            "gemm_bwd_pair": 2 * B * (pair(wc) + pair(wa)),
            # critic dQ/da chain (layers 2..L + the 10 action columns of layer 1) + the top layer's dgrad of both backward passes
            "gemm_dgrad": 2 * B * (sum(wc[1:]) + 10 * h1 + wc[L - 1] + wa[L - 1]),
            # the tails: wgrad(1) + wgrad(0), critic + actor
            "gemm_wgrad": 2 * B * (wc[1] + wc[0] + wa[1] + wa[0]),
        }
    return {
        # {actor_target, actor} and {critic_target, critic} layers 2..L, two layers per launch
        "gemm_fwd_lds_4x2": 2 * B * (2 * sum(wa[1:]) + 2 * sum(wc[1:])),
        # critic(s, mu(s)) layers 2..L
        "gemm_fwd_lds_2x2": 2 * B * sum(wc[1:]),
        # first layers (K = 58 / 68): 2 actor + 3 critic passes
        "gemm_fwd_direct": 2 * B * (2 * wa[0] + 3 * wc[0]),
        # dgrad+wgrad of layers 2..L: critic train + actor
        "gemm_bwd_pair": 2 * B * (2 * sum(wc[1:]) + 2 * sum(wa[1:])),
        # critic dQ/da chain: layers 2..L plus the 10 action columns of layer 1
        "gemm_dgrad": 2 * B * (sum(wc[1:]) + 10 * h1),
        # first-layer wgrads: critic + actor
        "gemm_wgrad": 2 * B * (wc[0] + wa[0]),
    }


def family_flops16(B, S, hidden):
    """Same algorithmic FLOPs, grouped by the fp16 learner's timing families (hgemm_nt launches)."""
    wa = tower_weights(S, hidden)
    wc = tower_weights(S + 10, hidden)
    h1 = hidden[0]
    return {
        "hgemm_fwd": 2 * B * (2 * sum(wa) + 3 * sum(wc)),
        "hgemm_dgrad": 2 * B * (sum(wc[1:]) + sum(wa[1:]) + sum(wc[1:]) + 10 * h1),
        "hgemm_wgrad": 2 * B * (sum(wc) + sum(wa)),
    }


MFMA_F16_PEAK_TF = 2500.0      # MI355X_MICROARCH.md: fp16/bf16 dense MFMA peak


KERNEL_NAMES = {"hgemm_fwd": "hgemm_nt<2,2>/<1,1> (forward epilogue)", "hgemm_dgrad": "hgemm_nt (dgrad epilogue)",
                "hgemm_wgrad": "hgemm_nt (wgrad epilogue)","gemm_fwd_lds_4x2": "gemm_fwd_lds<4,2,true,1>", "gemm_fwd_lds_2x2": "gemm_fwd_lds<2,2,true,2>",
                "gemm_fwd_direct": "gemm_fwd_direct<4,2>", "gemm_bwd_pair": "gemm_bwd_seq<true>",
                "gemm_dgrad": "gemm_dgrad_lds<1,1>", "gemm_wgrad": "gemm_wgrad_tail<1>"}


# kernel name (as rocprofv3 prints it, spaces removed) -> the learner's timing family
TRACE_FAMILY = (("gemm_fwd_lds<4,2,true,1>", "gemm_fwd_lds_4x2"), ("gemm_fwd_lds<4,2,true>", "gemm_fwd_lds_4x2"), ("gemm_fwd_lds<2,2,true", "gemm_fwd_lds_2x2"),
                ("gemm_fwd_lds<1,1,true", "gemm_fwd_lds_2x2"), ("gemm_fwd_direct", "gemm_fwd_direct"), ("gemm_bwd_seq", "gemm_bwd_pair"),
                ("gemm_bwd_pair_direct", "gemm_bwd_pair"), ("gemm_dgrad_lds", "gemm_dgrad"), ("gemm_dgrad_direct", "gemm_dgrad"), ("gemm_dgrad_narrow", "gemm_dgrad"), ("k_dqda_head_bwd", "gemm_dgrad"), ("k_dgrad_qtrain", "gemm_dgrad"),
                ("gemm_wgrad_tail", "gemm_wgrad"), ("gemm_wgrad_narrow", "gemm_wgrad"), ("k_adam_soft", "adam"))
