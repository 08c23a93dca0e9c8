# profiles/probes_r05.md section 2: the loss after the third L-BFGS step at 1024 x 1024 under 1e-6 perturbations of the start image (the 'lottery' table).
import json, os, sys, tempfile, torch
sys.path[:0] = ["/root/repo/maua-style_amd", "/root/repo/tests", "/root/repo"]
import config, models, optim, synth, plan, hip, engine
size = 1024
tmp = tempfile.mkdtemp()
wfile = os.path.join(tmp, "vgg19_synth.pth"); torch.save(synth.vgg19_state_dict(), wfile)
scaling = os.path.join(tmp, "scaling.json"); json.dump({"100000": {"gpu": "0", "multidevice": False}}, open(scaling, "w"))
args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", wfile, "--disable_check", "--scaling_args", scaling, "--optimizer", "lbfgs",
                        "--image_sizes", str(size), "--num_iters", "100", "--seed", "0", "--no_hist_match", "--no_grad_norm"])
args.hip_graph = False
optim.set_model_args(args, size)
content, style, init = synth.images(size)
for label, ov, eps in [("base", {"conv_x3p": "0"}, 0.0), ("base eps1e-6", {"conv_x3p": "0"}, 1e-6), ("base eps-1e-6", {"conv_x3p": "0"}, -1e-6), ("base eps3e-6", {"conv_x3p": "0"}, 3e-6),
                       ("x3 only", {"conv_x3p": "0", "conv_x3q": "0", "conv_x3w": "0"}, 0.0), ("x3p", {"conv_x3p": "64"}, 0.0), ("x3p eps1e-6", {"conv_x3p": "64"}, 1e-6), ("x3p eps-1e-6", {"conv_x3p": "64"}, -1e-6),
                       ("x3p max256 min256", {"conv_x3p": "256", "conv_x3p_max": "256"}, 0.0)]:
    plan.OVERRIDES.clear(); plan.OVERRIDES.update(ov)
    net, losses = models.load_model(args)
    optim.set_content_targets(net, content, args); optim.set_style_targets(net, [style], args)
    for m in losses: m.mode = "loss"
    opt = optim.PixelOptimizer(net, losses, init * (1.0 + eps), args)
    traj = []
    for _ in range(15):
        _, total = opt.step(); traj.append(float(total))
    print(f"{label:20s}", " ".join(f"{t:.4g}" for t in traj), flush=True)
