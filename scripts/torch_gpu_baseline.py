"""Calibration, not product and not an oracle: the same DDPG update (src/dqn.cpp:828-972 of the reference) written the
way one would with the stock framework on this GPU — PyTorch-ROCm fp32 modules (hipBLASLt / rocBLAS GEMMs), autograd,
fused capturable Adam, foreach clip and soft update — timed eagerly and as ONE captured HIP graph per update.
It answers "what does the vendor stack reach on this path on the same box", next to `python bench.py`.

  python scripts/torch_gpu_baseline.py [--minibatch 256] [--width 1024] [--steps 300] [--half]

Numerics follow the reference loosely (Adam epsilon placement, no inverting-gradient corner cases matter for time);
nothing here is compared with the library's results — parity lives in tests/.
"""
import argparse, json, os, time
import torch
import torch.nn as nn
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--minibatch", type=int, default=256)
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--state", type=int, default=58)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--replay", type=int, default=1_000_000)
ap.add_argument("--half", action="store_true", help="autocast fp16 GEMMs (fp32 master weights) — configs[4]'s arithmetic")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
B, S, W, L, NA, NP = args.minibatch, args.state, args.width, args.layers, 4, 6
NO = NA + NP


class Tower(nn.Module):
    def __init__(self, k_in, heads):
        super().__init__()
        dims = [k_in] + [W] * L
        self.body = nn.ModuleList(nn.Linear(dims[i], dims[i + 1]) for i in range(L))
        self.heads = nn.ModuleList(nn.Linear(W, h) for h in heads)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.01); nn.init.zeros_(m.bias)

    def forward(self, x):
        for l in self.body:
            x = F.leaky_relu(l(x), 0.01)
        return torch.cat([h(x) for h in self.heads], dim=1)


actor, critic = Tower(S, (NA, NP)).to(dev), Tower(S + NO, (1,)).to(dev)
actor_t, critic_t = Tower(S, (NA, NP)).to(dev), Tower(S + NO, (1,)).to(dev)
actor_t.load_state_dict(actor.state_dict()); critic_t.load_state_dict(critic.state_dict())
for p in list(actor_t.parameters()) + list(critic_t.parameters()):
    p.requires_grad_(False)
opt_a = torch.optim.Adam(actor.parameters(), lr=1e-5, betas=(0.95, 0.999), fused=True, capturable=True)
opt_c = torch.optim.Adam(critic.parameters(), lr=1e-3, betas=(0.95, 0.999), fused=True, capturable=True)

# device-resident replay ring (synthetic), sampled on the device like the library's sampler
N = args.replay
g = torch.Generator(device=dev); g.manual_seed(1)
R_s = torch.rand(N, S, device=dev, generator=g) * 2 - 1
R_a = torch.rand(N, NO, device=dev, generator=g) * 2 - 1
R_r = torch.rand(N, device=dev, generator=g) - 0.5
R_mc = torch.rand(N, device=dev, generator=g) - 0.5
R_sn = torch.rand(N, S, device=dev, generator=g) * 2 - 1
R_term = torch.rand(N, device=dev, generator=g) < 0.01
mn = torch.tensor([-1.] * 4 + [0., -180., -180., -180., 0., -180.], device=dev)
mx = torch.tensor([1.] * 4 + [100., 180., 180., 180., 100., 180.], device=dev)
gamma, beta, tau, clip = 0.99, 0.5, 0.001, 10.0
stats = torch.zeros(2, device=dev)


def update():
    idx = torch.randint(0, N, (B,), device=dev)
    s, a, r, mc, sn, term = R_s[idx], R_a[idx], R_r[idx], R_mc[idx], R_sn[idx], R_term[idx]
    with torch.autocast("cuda", dtype=torch.float16, enabled=args.half):
        with torch.no_grad():
            q_t = critic_t(torch.cat([sn, actor_t(sn).float()], 1))[:, 0].float()
            y = beta * mc + (1 - beta) * torch.where(term, r, r + gamma * q_t)
        q = critic(torch.cat([s, a], 1))[:, 0].float()
        loss = ((q - y) ** 2).sum() / (2 * B)
    opt_c.zero_grad(set_to_none=False)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(critic.parameters(), clip, foreach=True)
    opt_c.step()
    with torch.autocast("cuda", dtype=torch.float16, enabled=args.half):
        mu = actor(s).float()
        mu_d = mu.detach().requires_grad_(True)
        q2 = critic(torch.cat([s, mu_d], 1))[:, 0].float()
    (dq,) = torch.autograd.grad(-q2.sum(), mu_d)
    out = mu.detach()
    inv = torch.where(dq < 0, dq * (mx - out) / (mx - mn), torch.where(dq > 0, dq * (out - mn) / (mx - mn), dq))
    opt_a.zero_grad(set_to_none=False)
    mu.backward(inv)
    torch.nn.utils.clip_grad_norm_(actor.parameters(), clip, foreach=True)
    opt_a.step()
    with torch.no_grad():
        for net, tgt in ((critic, critic_t), (actor, actor_t)):
            ps, ts = list(net.parameters()), list(tgt.parameters())
            torch._foreach_mul_(ts, 1 - tau); torch._foreach_add_(ts, ps, alpha=tau)
        stats[0] = loss.detach(); stats[1] = q2.detach().sum() / B


def timed(fn, steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for _ in range(5):
    update()
eager_ms = timed(update, args.steps)
graph_ms = None
try:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            update()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        update()
    for _ in range(5):
        gr.replay()
    graph_ms = timed(gr.replay, args.steps)
except Exception as e:                                   # capture can fail on an op that syncs: report, keep the eager number
    print("graph capture failed:", repr(e)[:300])
rec = {"what": "PyTorch-ROCm %s update, same path, same GPU (calibration)" % ("fp16-autocast" if args.half else "fp32"),
       "torch": torch.__version__, "minibatch": B, "tower": "%dx%d" % (L, W), "state": S,
       "eager_ms_per_update": round(eager_ms, 4), "graph_ms_per_update": None if graph_ms is None else round(graph_ms, 4),
       "graph_updates_per_s": None if graph_ms is None else round(1e3 / graph_ms, 1), "loss": float(stats[0])}
print(json.dumps(rec))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/torch_gpu_baseline_%s_b%d.json" % ("fp16" if args.half else "fp32", B), "w") as f:
    json.dump(rec, f, indent=1)
