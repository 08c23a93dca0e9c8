"""Where the non-iteration time of optim.optimize goes at the scales of BASELINE config 3 (one net, growing sizes): wall clock of the
whole call against its iteration loop alone, synchronised.
    python tools/probe_setup_time.py [sizes ...]"""
import os, sys, time, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd"), os.path.join(REPO, "tests")]
import torch
from conftest import product_args
import synth, optim, models

sizes = [int(v) for v in sys.argv[1:]] or [256, 512, 1024, 2048]
d = tempfile.mkdtemp()
wf = {"vgg19": os.path.join(d, "vgg19_synth.pth"), "nin": os.path.join(d, "nin_synth.pth")}
torch.save(synth.vgg19_state_dict(), wf["vgg19"])


def now():
    torch.cuda.synchronize()
    return time.perf_counter()


loop = {}
orig = optim._run_iterations


def timed_loop(opt, n, args, *a, **k):
    t0 = now()
    orig(opt, n, args, *a, **k)
    loop["s"] = now() - t0


optim._run_iterations = timed_loop
net = losses = None
N = 60
for S in sizes:
    opt_name = "adam" if S >= 2048 else "lbfgs"
    args = product_args(wf, optimizer=opt_name, S=S, N=N)
    content, style, init = synth.images(S)
    if net is None:
        optim.set_model_args(args, S)
        t0 = now(); net, losses = models.load_model(args); t_load = now() - t0
        print(f"load_model {t_load * 1e3:.0f} ms")
    for rep in range(2):  # second call at the same size: what a warm allocator / cached banks / captured graph leave
        t0 = now()
        out = optim.optimize(content, [style], init.clone(), N, args, net, losses, keep_on_device=True)
        total = now() - t0
        print(f"{S}x{S} {opt_name} call {rep}: optimize {total * 1e3:7.1f} ms, of which iteration loop {loop['s'] * 1e3:7.1f} ms "
              f"({loop['s'] / N * 1e3:.3f} ms/iter incl. graph capture) -> setup {1e3 * (total - loop['s']):6.1f} ms", flush=True)
