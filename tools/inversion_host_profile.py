"""The inversion forward of bench.py (pass #1, texture head + FiLM, pass #2, decoder to 1024^2) run back to back: wall time per
forward, and with `--profile` a cProfile of the HOST side (the GPU needs ~1.25 ms per forward; the Python that issues its ~30
launches has to stay under that).   python tools/inversion_host_profile.py [--profile] [n]"""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import e3dge_amd  # noqa: F401,E402
from e3dge_amd import synthetic as syn  # noqa: E402
from e3dge_amd.camera_utils import generate_camera_params  # noqa: E402
from e3dge_amd.stylesdf_model import G_pred_latents  # noqa: E402

prof = "--profile" in sys.argv
nums = [a for a in sys.argv[1:] if a.isdigit()]
n = int(nums[0]) if nums else 50
dev = "cuda:0"
gl = G_pred_latents(syn.model_opt(), syn.rendering_opt(N_samples=24, enable_local_model=True, L_pred_tex_modulations=True),
                    full_pipeline=True)
syn.load_synthetic(gl)
gl = gl.to(dev).eval()
gl.requires_grad_(False)
p1, f1, n1, fa1, _ = generate_camera_params(64, dev, locations=torch.zeros(1, 2, device=dev))
feats = syn.synthetic_local_feats(1, 64, 24, device=dev)
w_r, w_d = syn.synthetic_inputs(1, seed=1, device=dev)


def inversion():
    gl([w_r, w_d], p1, f1, n1, fa1, input_is_latent=True, sample_with_renderer=True)
    return gl([w_r, w_d], p1, f1, n1, fa1, input_is_latent=True, randomize_noise=False, local_data_batch={'feats': feats})


with torch.no_grad():
    for _ in range(5):
        inversion()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        inversion()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"inversion forward x{n}: {1e3 * t_all / n:.3f} ms per forward; the host had issued everything after {1e3 * t_issue / n:.3f} ms per forward")
    if prof:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            inversion()
        pr.disable()
        torch.cuda.synchronize()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
        print("\n".join(l[:160] for l in s.getvalue().splitlines()[:60]))
