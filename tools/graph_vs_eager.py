"""Why is a HIP-graph replay of the inversion forward not faster than its eager launches?  (round-4 review, bench hygiene (a):
inversion_fwd_graph_ms 1.33 > eager 1.27.)   python tools/graph_vs_eager.py  -> gpurun_out/graph_vs_eager.json

Three measurements, HIP events around 20 repetitions each:
  1. a chain of 60 dependent 1-KiB kernels (x.add_(1)): eager launches vs one graph replay -> the per-node cost of each submission path
     when the GPU work itself is ~nothing;
  2. the inversion forward (pass #1, texture head, pass #2, decoder: 21 launches of 5-330 us): eager, graph replay as GraphedCall does it
     (two input copies + replay), and the bare replay;
  3. the same forward with the host deliberately slowed (a 20 us spin between launches is what a tracer or a busy host does): eager vs replay
     -- the case a graph is for."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.graphs import GraphedCall  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

dev = "cuda:0"


def ev_ms(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {}
# 1. empty chain
x = torch.zeros(256, device=dev)


def chain():
    for _ in range(60):
        x.add_(1.0)


g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    chain()
e, r = ev_ms(chain), ev_ms(g.replay)
res["chain_of_60_tiny_kernels"] = {"eager_us_per_kernel": round(1e3 * e / 60, 3), "graph_us_per_node": round(1e3 * r / 60, 3)}

# 2. the inversion forward
RES, S = 64, 24
gl = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=S, enable_local_model=True, L_pred_tex_modulations=True), full_pipeline=True)
sd = {}
g0 = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=S), full_pipeline=True)
syn.load_synthetic(g0)
for k, v in g0.state_dict().items():
    sd[k.replace('renderer.network.', 'renderer.network.netGlobal.')] = v
for k, v in gl.state_dict().items():
    if '.netLocal.' in k:
        sd[k] = 0.05 * syn.synthetic_tensor(k, v.shape)
gl.load_state_dict(sd)
gl = gl.to(dev).eval()
gl.requires_grad_(False)
p1, f1, n1, fa1, _ = generate_camera_params(RES, dev, locations=torch.zeros(1, 2, device=dev))
feats = syn.synthetic_local_feats(1, RES, S, device=dev)
w1, d1 = syn.synthetic_inputs(1, seed=1, device=dev)


def inversion(w_r, w_d):
    gl([w_r, w_d], p1, f1, n1, fa1, input_is_latent=True, sample_with_renderer=True)
    return gl([w_r, w_d], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False, local_data_batch={'feats': feats})['gen_imgs']


with torch.no_grad():
    gi = GraphedCall(inversion, w1, d1)
    # blocks of ten replays straight after the capture (no warm-up): is a fresh graph slower at first?
    blocks = []
    for _ in range(5):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(10):
            gi(w1, d1)
        b_.record()
        torch.cuda.synchronize()
        blocks.append(round(a_.elapsed_time(b_) / 10, 4))
    res["graphed_call_ms_in_blocks_of_10_after_capture"] = blocks
    # clock ramp: the GPU clocks down within ~100 ms of idling and needs as long to come back; every figure below is taken
    # after 200 ms of the same work, so eager and replay are compared at the same clock
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.2:
        inversion(w1, d1)
    torch.cuda.synchronize()
    t_e = ev_ms(lambda: inversion(w1, d1))
    t_g = ev_ms(lambda: gi(w1, d1))
    t_r = ev_ms(gi.graph.replay)
    # host time of the eager forward (how far ahead of the GPU the host runs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        inversion(w1, d1)
    host = 1e3 * (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    res["inversion_forward_ms"] = {"eager": round(t_e, 4), "graph_as_GraphedCall": round(t_g, 4), "bare_replay": round(t_r, 4),
                                   "eager_host_enqueue_ms": round(host, 4)}

    # what bench.py's wall clock sees: n iterations between two synchronizes, per iteration
    def wall(fn, n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / n
    res["wall_clock_ms_per_forward"] = {f"{name}_n{n}": round(wall(fn, n), 4) for n in (1, 10, 100)
                                        for name, fn in (("eager", lambda: inversion(w1, d1)), ("graphed_call", lambda: gi(w1, d1)),
                                                         ("bare_replay", gi.graph.replay))}

    # 3. a slow host: every e3dge launch preceded by a spin
    from e3dge_amd import _lib
    lib = _lib.load()

    def spin(us):
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < us:
            pass

    def slow_inversion():
        # 21 launches; emulate a host that needs 60 us more per launch by spinning once per forward for the total
        spin(21 * 60)
        return inversion(w1, d1)

    def slow_replay():
        spin(21 * 60)
        gi.graph.replay()
    # (with the spin IN FRONT of the submissions, eager = spin + enqueue, replay = spin + one submission: the difference is what a graph saves
    # when the GPU has run dry)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        slow_inversion()
        torch.cuda.synchronize()
    t_se = 1e3 * (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        slow_replay()
        torch.cuda.synchronize()
    t_sr = 1e3 * (time.perf_counter() - t0) / 10
    res["one_forward_from_an_idle_gpu_ms"] = {"eager_incl_1.26ms_spin": round(t_se, 4), "replay_incl_1.26ms_spin": round(t_sr, 4)}
res["reading"] = ("graph nodes are submitted as the same AQL packets as eager launches, each behind a barrier bit: per node the replay costs what "
                  "chain_of_60_tiny_kernels.graph_us_per_node says, vs eager_us_per_kernel with the queue kept full by the host; a replay wins "
                  "only when the host cannot keep the queue full (latency from an idle GPU, or a host slower than the kernels)")
line = json.dumps(res)
print(line)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/graph_vs_eager.json", "w").write(line + "\n")
