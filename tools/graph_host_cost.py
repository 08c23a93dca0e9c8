"""Host cost of one hipGraph replay of an L-BFGS iteration vs the GPU time of that iteration, per image size: is a small size bound by
hipGraphLaunch on the host (per-node cost) rather than by its kernels?   python tools/graph_host_cost.py [sizes...]
Prints, per size: graph nodes (kernel launches) per iteration, host seconds per replay() call with the queue kept short (sync every
call: host + GPU serial), with the queue free-running (what bench.py times), and host-only (time for the calls to RETURN, GPU behind)."""
import json, os, sys, tempfile, time
import torch
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "maua-style_amd"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")]
import config, models, optim, synth

def build(size, history=100):
    tmp = tempfile.mkdtemp(prefix="maua_probe_")
    wfile = os.path.join(tmp, "vgg19_synth.pth")
    torch.save(synth.vgg19_state_dict(), wfile)
    scaling = os.path.join(tmp, "scaling.json")
    json.dump({"100000": {"gpu": "0", "multidevice": False}}, open(scaling, "w"))
    args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", wfile, "--disable_check", "--scaling_args", scaling,
                            "--optimizer", "lbfgs", "--image_sizes", str(size), "--num_iters", "1000", "--seed", "0", "--no_hist_match",
                            "--lbfgs_num_correction", str(history)])
    args.hip_graph = True
    optim.set_model_args(args, size)
    net, losses = models.load_model(args)
    content, style, init = synth.images(size)
    optim.set_content_targets(net, content, args)
    optim.set_style_targets(net, [style], args)
    for m in losses:
        m.mode = "loss"
    opt = optim.PixelOptimizer(net, losses, init, args)
    for _ in range(history + 5):
        opt.step()
    torch.cuda.synchronize()
    return opt

for size in [int(v) for v in sys.argv[1:]] or [256, 512]:
    opt = build(size)
    g = opt._graph
    K = 300
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K):
        g.replay()
    t_ret = time.perf_counter() - t
    torch.cuda.synchronize(); t_free = time.perf_counter() - t
    t = time.perf_counter()
    for _ in range(K):
        g.replay(); torch.cuda.synchronize()
    t_sync = time.perf_counter() - t
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"size {size}: per replay  calls return in {t_ret / K * 1e6:.0f} us   free-running {t_free / K * 1e6:.0f} us   with a sync per call {t_sync / K * 1e6:.0f} us   GPU events {e0.elapsed_time(e1) / K * 1e3:.0f} us")
